"""What a small RCCL collective costs on this box with a world of ONE (no data moves: the launch, the kernel, the stream hand-offs) -- a lower
bound for the fixed cost per collective that tools/predict_scaling.py assumes (20 us)."""
import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541"); os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
st = torch.cuda.Stream()
for name, t, op in (("int64 x 7600 (a partitioned pass's exchange at 483k structures), SUM", torch.zeros(7600, dtype=torch.int64, device="cuda"), dist.ReduceOp.SUM),
                    ("int32 x 483472 (best[] of a row-tile pass), MIN", torch.zeros(483472, dtype=torch.int32, device="cuda"), dist.ReduceOp.MIN),
                    ("uint8 x 1000000 (hybrid front's mask), SUM", torch.zeros(1_000_000, dtype=torch.uint8, device="cuda"), dist.ReduceOp.SUM)):
    with torch.cuda.stream(st):
        for _ in range(20): dist.all_reduce(t, op=op)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st)
        for _ in range(200): dist.all_reduce(t, op=op)
        b.record(st); b.synchronize()
        gpu_us = a.elapsed_time(b) / 200 * 1e3
        t0 = time.perf_counter()
        for _ in range(200): dist.all_reduce(t, op=op)
        host_us = (time.perf_counter() - t0) / 200 * 1e6
        torch.cuda.synchronize()
    print(f"{name}: {gpu_us:.1f} us per call on the stream (back to back), {host_us:.1f} us of host time per call")
dist.destroy_process_group()
