#!/usr/bin/env python3
"""A/B timing of builds of EPIT's epipolar attention (lfsr_window_attn_fwd at the EPIT geometry, both passes) inside ONE process, variants interleaved.
usage: python tools/attn_ab.py tag=lib.so [tag=lib.so ...]    env: AB_B (default 8), AB_ROUNDS, AB_REPS"""
import ctypes as C, os, sys, statistics
import torch
c_p, c_i, c_ll = C.c_void_p, C.c_int, C.c_longlong


def bind(path):
    lib = C.CDLL(os.path.abspath(path))
    lib.lfsr_window_attn_fwd.restype = c_i
    lib.lfsr_window_attn_fwd.argtypes = [c_p, c_i, c_i, c_p, c_i, c_i, c_p, c_i, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_ll, c_ll, c_ll, c_i, c_i, c_ll, c_ll, c_i, c_i, c_i, c_i, c_i, c_p]
    return lib


def main():
    libs = [(a.split("=", 1)[0], bind(a.split("=", 1)[1])) for a in sys.argv[1:]]
    B, A, h, w, E, NH = int(os.environ.get("AB_B", "8")), 5, 32, 32, 128, 8
    rounds, reps = int(os.environ.get("AB_ROUNDS", "6")), int(os.environ.get("AB_REPS", "20"))
    npix, HW = B * A * A * h * w, h * w
    g = torch.Generator(device="cuda").manual_seed(5)
    qk = torch.randn(npix, 256, device="cuda", generator=g); v = torch.randn(npix, 128, device="cuda", generator=g)
    o = torch.empty(npix, 128, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    geoms = {"H": (B, A, w, A * A * HW, HW, 1, A, h, A * HW, w), "V": (B, A, h, A * A * HW, A * HW, w, A, w, HW, 1)}

    def run(lib, key):
        rc = lib.lfsr_window_attn_fwd(qk.data_ptr(), 256, 0, qk.data_ptr(), 256, 128, v.data_ptr(), 128, 0, o.data_ptr(), 128, 0, NH, E // NH, *geoms[key], A, A, 5, 6, 0, st)
        assert rc == 0, rc
    ref = {}
    for t, lib in libs:
        for key in geoms:
            o.zero_(); run(lib, key); torch.cuda.synchronize()
            if key not in ref: ref[key] = o.clone()
            else: print(f"check {key} {t:8s} max|d vs {libs[0][0]}| = {float((o - ref[key]).abs().max()):.3e}", flush=True)
    for _ in range(3):
        for t, lib in libs:
            for _ in range(10): run(lib, "H")
    torch.cuda.synchronize()
    times = {(t, k): [] for t, _ in libs for k in geoms}
    for rd in range(rounds):
        for key in geoms:
            for t, lib in (libs if rd % 2 == 0 else libs[::-1]):
                for _ in range(3): run(lib, key)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps): run(lib, key)
                e1.record(); torch.cuda.synchronize()
                times[(t, key)].append(e0.elapsed_time(e1) * 1e3 / reps)
    for t, _ in libs:
        print(f"{t:10s} " + "  ".join(f"{k}: {statistics.median(times[(t, k)]):7.1f} ({min(times[(t, k)]):6.1f}..{max(times[(t, k)]):6.1f}) us" for k in geoms), flush=True)


if __name__ == "__main__":
    main()
