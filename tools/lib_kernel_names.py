"""Which kernels does the vendor GEMM library pick for the step's GEMM shapes?  Run under
`rocprofv3 --kernel-trace --stats` and read the Tensile kernel names (macro tile MT, matrix instruction MI, depthU,
prefetch depths, stream-K flags): intelligence for the own kernels' tile / instruction choices, not a dependency.

usage: rocprofv3 --kernel-trace --stats -d gpurun_out/libnames -- python3 tools/lib_kernel_names.py"""
import torch

dev = torch.device("cuda")
shapes = [(8192, 8192, 8192), (4096, 4096, 4096), (80128, 768, 2048), (40064, 1536, 768), (160256, 1024, 512),
          (9984, 4096, 512), (9984, 512, 2048), (9984, 512, 4096), (10016, 768, 3072), (9984, 160, 2048)]
for dt in (torch.float16,):
    for M, N, K in shapes:
        a = torch.randn(M, K, device=dev).to(dt)
        w = torch.randn(N, K, device=dev).to(dt)
        for _ in range(3):
            c = torch.nn.functional.linear(a, w)          # NT: y = a w^T, the conv stack's forward form
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            c = torch.nn.functional.linear(a, w)
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 10
        print(f"NT {M}x{N}x{K} {dt}: {ms * 1e3:.1f} us  {2.0 * M * N * K / ms / 1e9:.0f} TFLOP/s", flush=True)
        # TN: dW = dz^T x (weight gradient form)
        dz = torch.randn(M, N, device=dev).to(dt)
        for _ in range(3):
            g = dz.t() @ a
        torch.cuda.synchronize()
        s.record()
        for _ in range(10):
            g = dz.t() @ a
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 10
        print(f"TN {M}x{N}x{K} {dt}: {ms * 1e3:.1f} us  {2.0 * M * N * K / ms / 1e9:.0f} TFLOP/s", flush=True)
        del a, w, c, dz, g
