#!/usr/bin/env python3
"""Where the stock PyTorch-ROCm step (bench.py's comparators.torch_rocm) spends its time: top operators by device time for
each variant (fp32 NCHW, bf16 autocast NCHW, bf16 autocast channels_last).  Diagnostic only."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import fixtures as fx  # noqa: E402
from oracle.torch_step import TorchStep  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    dev = torch.device("cuda:0")
    torch.backends.cudnn.benchmark = True
    P, Bf = fx.model_state(6, "concat_dgl")
    g = torch.Generator(device="cpu").manual_seed(1)
    data = (torch.randn(B, 257, 188, generator=g).to(dev), torch.randn(B, 3, 3, 224, 224, generator=g).to(dev),
            torch.randint(0, 6, (B,), generator=g).to(dev))
    for tag, ac, cl in (("fp32_nchw", None, False), ("bf16_nchw", torch.bfloat16, False), ("bf16_channels_last", torch.bfloat16, True),
                        ("fp32_channels_last", None, True)):
        ts = TorchStep(P, Bf, device=dev, autocast=ac, channels_last=cl)
        for _ in range(4):
            ts.train_step(*data, 4.0, 2e-3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            ts.train_step(*data, 4.0, 2e-3)
        torch.cuda.synchronize()
        print(f"== {tag}: {(time.perf_counter() - t0) / 3 * 1e3:.1f} ms / step", flush=True)
        from torch.profiler import ProfilerActivity, profile

        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
            ts.train_step(*data, 4.0, 2e-3)
            torch.cuda.synchronize()
        print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=70), flush=True)
        del ts
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
