"""Device time of one Sinkhorn solve (tt_sinkhorn: init + iterations + output launch), measured on a hipGraph replay of the launch sequence -
a loop of eager calls measures the HOST's launch rate (12 launches of ~5 us kernels) - for one or more library builds, interleaved.
    python tools/sk_time.py [B] [K] [iters] [label=lib.so ...]      (default: the in-tree library)"""
import ctypes as C, os, statistics, sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import timetuning_amd  # noqa: F401  (the hipGraph flag)

args = [a for a in sys.argv[1:] if "=" not in a]
libs = [a.split("=", 1) for a in sys.argv[1:] if "=" in a] or [("tree", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "timetuning_amd", "libtimetuning_hip.so"))]
B = int(args[0]) if args else 6272
K = int(args[1]) if len(args) > 1 else 200
iters = int(args[2]) if len(args) > 2 else 10
torch.manual_seed(0)
sc = (torch.nn.functional.normalize(torch.randn(B, 256, device="cuda"), dim=1) @ torch.nn.functional.normalize(torch.randn(K, 256, device="cuda"), dim=1).t()).contiguous()
vp = C.c_void_p
runs = {}
for name, path in libs:
    lib = C.CDLL(path)
    lib.tt_sinkhorn_workspace_bytes.restype = C.c_size_t
    lib.tt_sinkhorn_workspace_bytes.argtypes = [C.c_int, C.c_int]
    lib.tt_sinkhorn.restype = C.c_int
    lib.tt_sinkhorn.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, vp, C.c_size_t, vp]
    nb = lib.tt_sinkhorn_workspace_bytes(B, K)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    q = torch.empty(B, K, device="cuda")

    def call(lib=lib, ws=ws, q=q, nb=nb):
        rc = lib.tt_sinkhorn(sc.data_ptr(), q.data_ptr(), B, K, 0, B, 0.05, iters, ws.data_ptr(), nb, torch.cuda.current_stream().cuda_stream)
        assert rc == 0

    call(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        call()
    runs[name] = (g, q, call, [], [])
for rd in range(12):
    for name, (g, q, call, tg, te) in runs.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): g.replay()
        e1.record(); torch.cuda.synchronize()
        if rd >= 2: tg.append(e0.elapsed_time(e1) * 100)
        e0.record()
        for _ in range(10): call()
        e1.record(); torch.cuda.synchronize()
        if rd >= 2: te.append(e0.elapsed_time(e1) * 100)
ref = None
for name, (g, q, call, tg, te) in runs.items():
    g.replay(); torch.cuda.synchronize()
    d = "" if ref is None else f"  max |q - q_first| {float((q - ref).abs().max()):.2e}"
    ref = q.clone() if ref is None else ref
    us = statistics.median(tg)
    print(f"{name:8s} B={B} K={K} iters={iters}: graph replay {us:7.1f} us per solve ({iters / us * 1e6 / 1e3:6.1f} k iters/s) | eager loop {statistics.median(te):7.1f} us{d}")
