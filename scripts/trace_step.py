"""Kernel-by-kernel listing of ONE eager training step (torch.profiler): which host op launched each
kernel, in launch order, with its device time.  Used to find the small launches a step still makes
between the hand-written kernels.   usage: python scripts/trace_step.py [--model dcn] [--batch 4096]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import bench
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="dcn")
    ap.add_argument("--batch", type=int, default=4096)
    a = ap.parse_args()
    sys.argv = ["bench.py", "--model", a.model, "--batch", str(a.batch), "--no-tunable"]
    args = bench.parse_args()
    dev = torch.device("cuda", 0)
    est, spec, feats, labels, workload = bench.build_estimator(args, dev, 0, 1)
    for _ in range(3):
        est.train_step(feats, labels)
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        est.train_step(feats, labels)
        torch.cuda.synchronize()
    evs = prof.events()
    kern = sorted([e for e in evs if e.device_type == torch.autograd.DeviceType.CUDA], key=lambda e: e.time_range.start)
    print(f"# {workload}: {len(kern)} device activities in one eager step")
    total = 0.0
    for k in kern:
        name = k.name
        dur = k.time_range.elapsed_us()
        total += dur
        print(f"{dur:8.1f} us  {name[:150]}")
    print(f"# sum of device time {total:.1f} us")
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=60, max_name_column_width=90))


if __name__ == "__main__":
    main()
