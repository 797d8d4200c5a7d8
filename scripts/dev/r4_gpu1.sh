#!/bin/bash
# round-4 GPU call 1: stand-alone reproducer, overlap soak, packed-build variants, full GPU suite, bench
mkdir -p gpurun_out/r4a
O=$PWD/gpurun_out/r4a
( cd scripts/ubench; for b in lds_pk_overlap lds_pk_overlap_NOP lds_pk_overlap_B64 lds_pk_overlap_MOV; do echo "== $b"; timeout 300 ./$b 120 96; done ) > $O/ubench.txt 2>&1
timeout 1200 python -m pytest tests/test_gpu_overlap.py -q -s > $O/overlap.txt 2>&1
for v in nop mov b64; do echo "== variant $v"; PSCV_LIB=$PWD/scripts/dev/libpscv_pk_$v.so timeout 400 python -m pytest tests/test_gpu_overlap.py -k packed -s -q; done > $O/variants.txt 2>&1
timeout 1200 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_overlap.py > $O/gpu_tests.txt 2>&1
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
tail -3 $O/ubench.txt $O/overlap.txt $O/variants.txt $O/gpu_tests.txt; head -c 600 $O/bench.json
