"""Which torch.distributed calls does gloo accept for CUDA tensors (two ranks on one GPU, as in tests/test_gpu_dist.py), and are
they ordered with the CUDA stream?  Every input is produced by GPU work queued right before the call (a chain of large
element-wise kernels), every output is consumed right after it; values are checked."""
import os
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def w(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = "cuda" if torch.cuda.is_available() else "cpu"
    n = 32 * 1024 * 1024

    def fresh(val):
        x = torch.zeros(n, dtype=torch.float16, device=dev)
        for _ in range(20):                 # ~20 kernels queued: the collective must wait for them
            x = x + val / 20
        return x
    results = {}

    def case(name, fn):
        try:
            results[name] = fn()
        except Exception as e:
            results[name] = f"FAIL {type(e).__name__}: {str(e)[:80]}"

    def rs():
        out = torch.empty(n // 2, dtype=torch.float16, device=dev)
        dist.reduce_scatter_tensor(out, fresh(rank + 1.0))
        return float((out.float() - 3.0).abs().max())

    def ar():
        x = fresh(rank + 1.0)
        dist.all_reduce(x)
        return float((x.float() - 3.0).abs().max())

    def ag():
        bufs = [torch.empty(n, dtype=torch.float16, device=dev) for _ in range(2)]
        dist.all_gather(bufs, fresh(rank + 1.0))
        return max(float((bufs[r].float() - (r + 1.0)).abs().max()) for r in range(2))

    def p2p():
        rbuf = torch.empty(n, dtype=torch.float16, device=dev)
        x = fresh(rank + 1.0)
        for r in dist.batch_isend_irecv([dist.P2POp(dist.isend, x, 1 - rank), dist.P2POp(dist.irecv, rbuf, 1 - rank)]):
            r.wait()
        return float((rbuf.float() - (2.0 - rank)).abs().max())
    case("reduce_scatter_tensor", rs)
    case("all_reduce", ar)
    case("all_gather", ag)
    case("batch_isend_irecv", p2p)
    if rank == 0:
        for k, v in results.items():
            print(f"{k}: max abs error {v}", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    mp.spawn(w, args=(2, 29611), nprocs=2)
