#!/usr/bin/env python3
"""Re-flow a markdown file to <= WIDTH columns: paragraphs and bullets are wrapped (hanging indent kept), table rows whose cells are
prose (any line over WIDTH) become bullet lists "- **first cell** -- second cell; ..." (a table with one 2 000-character cell is not a
table); code fences, headings and short tables are left alone.  Usage: python scripts/dev/rewrap_md.py IN.md OUT.md [WIDTH]"""
import re
import sys
import textwrap


def wrap(text, width, first="", rest=""):
    return textwrap.fill(text, width=width, initial_indent=first, subsequent_indent=rest, break_long_words=False, break_on_hyphens=False)


def main():
    src, dst = sys.argv[1], sys.argv[2]
    width = int(sys.argv[3]) if len(sys.argv) > 3 else 110
    lines = open(src).read().split("\n")
    out, i, fence = [], 0, False
    while i < len(lines):
        l = lines[i]
        if l.lstrip().startswith("```"):
            fence = not fence
            out.append(l); i += 1; continue
        if fence or not l.strip() or l.startswith("#"):
            out.append(l); i += 1; continue
        if l.lstrip().startswith("|"):
            j = i
            while j < len(lines) and lines[j].lstrip().startswith("|"):
                j += 1
            block = lines[i:j]
            if max(len(b) for b in block) <= width + 30:
                out.extend(block)
            else:
                rows = [[c.strip() for c in b.strip().strip("|").split("|")] for b in block]
                header = rows[0]
                body = [r for r in rows[1:] if not all(re.fullmatch(r":?-+:?", c) for c in r if c)]
                out.append(wrap("(table: " + " / ".join(h for h in header if h) + ")", width))
                out.append("")
                for r in body:
                    cells = [c for c in r]
                    head = cells[0] if cells else ""
                    rest = []
                    for h, c in zip(header[1:], cells[1:]):
                        if c:
                            rest.append(f"*{h}*: {c}" if h else c)
                    out.append(wrap(f"- **{head}** -- " + "; ".join(rest), width, "", "  "))
                out.append("")
            i = j
            continue
        m = re.match(r"(\s*)([*+-]|\d+\.)\s+", l)
        if m:
            ind = m.group(1)
            lead = l[:m.end()]
            text = l[m.end():]
            j = i + 1
            while j < len(lines) and lines[j].strip() and not re.match(r"\s*([*+-]|\d+\.)\s+", lines[j]) and not lines[j].startswith("#") \
                    and not lines[j].lstrip().startswith("|") and not lines[j].lstrip().startswith("```") and lines[j].startswith(ind + " "):
                text += " " + lines[j].strip(); j += 1
            out.append(wrap(text, width, lead, " " * len(lead)))
            i = j
            continue
        j = i
        text = []
        while j < len(lines) and lines[j].strip() and not lines[j].startswith("#") and not lines[j].lstrip().startswith("|") \
                and not lines[j].lstrip().startswith("```") and not re.match(r"\s*([*+-]|\d+\.)\s+", lines[j]):
            text.append(lines[j].strip()); j += 1
        out.append(wrap(" ".join(text), width))
        i = j
    open(dst, "w").write("\n".join(out))


if __name__ == "__main__":
    main()
