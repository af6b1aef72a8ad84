#!/usr/bin/env python3
"""A/B timing of builds of the 3x3 conv kernel inside ONE process, variants interleaved round-robin (device-to-device and clock-ramp
spread makes figures of different processes incomparable).
usage: python tools/conv_ab.py tag=lib.so [tag=lib.so ...]      (the first one is the reference for the bit comparison)
env: AB_ROUNDS (default 6), AB_REPS (default 20), AB_GEOMS (default "800:n,800:y,200:n"), AB_NW (default 1): that many different packed weights used in turn
(16 x 590 KB do not fit an XCD's L2: every launch then streams its U fragments cold, as the convs of a forward do)"""
import ctypes as C, os, sys, statistics
import torch

c_p, c_i, c_f = C.c_void_p, C.c_int, C.c_float


def bind(path):
    lib = C.CDLL(os.path.abspath(path))
    lib.lfsr_conv3x3_fwd.restype = c_i
    lib.lfsr_conv3x3_fwd.argtypes = [c_p, c_i, c_i, c_p, c_p, c_i, c_i, c_p, c_i, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_f, c_p]
    lib.lfsr_packed_weight_floats.restype = C.c_longlong
    lib.lfsr_packed_weight_floats.argtypes = [c_i, c_i, c_i]
    lib.lfsr_pack_conv_weight.restype = c_i
    lib.lfsr_pack_conv_weight.argtypes = [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p]
    return lib


def main():
    specs = [a.split("=", 1) for a in sys.argv[1:]]
    libs = [(t, bind(p)) for t, p in specs]
    rounds, reps = int(os.environ.get("AB_ROUNDS", "6")), int(os.environ.get("AB_REPS", "20"))
    geoms = [(int(g.split(":")[0]), g.split(":")[1] == "y") for g in os.environ.get("AB_GEOMS", "800:n,800:y,200:n").split(",")]
    h = w = 32
    gen = torch.Generator(device="cuda").manual_seed(3)
    nmax = max(n for n, _ in geoms)
    x = torch.randn(nmax * h * w, 64, device="cuda", generator=gen)
    r = torch.randn(nmax * h * w, 64, device="cuda", generator=gen)
    wt = torch.randn(64, 64, 3, 3, device="cuda", generator=gen) * 0.05
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    packs = {}
    nw = int(os.environ.get("AB_NW", "1"))
    for t, lib in libs:
        n = lib.lfsr_packed_weight_floats(64, 64, 9)
        packs[t] = []
        for _ in range(nw):
            wp = torch.empty(n, device="cuda")
            assert lib.lfsr_pack_conv_weight(wt.data_ptr(), wp.data_ptr(), 64, 64, 9, 0, 0, st) == 0
            packs[t].append(wp)
    cnt = [0]
    y = torch.empty(nmax * h * w, 64, device="cuda")

    def run(lib, t, n_img, res):
        cnt[0] += 1
        rc = lib.lfsr_conv3x3_fwd(x.data_ptr(), 64, 0, packs[t][cnt[0] % nw].data_ptr(), y.data_ptr(), 64, 0, r.data_ptr() if res else None, 64 if res else 0, 0,
                                  None, 0, 0, n_img, h, w, 0.1, st)
        assert rc == 0, rc

    # correctness against the first library
    for n_img, res in geoms:
        ref = None
        for t, lib in libs:
            y.zero_(); run(lib, t, n_img, res); torch.cuda.synchronize()
            out = y[: n_img * h * w].clone()
            if ref is None: ref = out
            else:
                d = float((out - ref).abs().max())
                print(f"check n_img={n_img} res={'y' if res else 'n'} {t:10s} max|d vs {libs[0][0]}| = {d:.3e}{'  (bit-equal)' if d == 0 else ''}", flush=True)
    # warm-up: clocks settle only after ~20 ms of work
    for _ in range(3):
        for t, lib in libs:
            for _ in range(10): run(lib, t, nmax, False)
    torch.cuda.synchronize()
    times = {(t, g): [] for t, _ in libs for g in geoms}
    for rd in range(rounds):
        order = libs if rd % 2 == 0 else libs[::-1]
        for g in geoms:
            for t, lib in order:
                for _ in range(3): run(lib, t, g[0], g[1])
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps): run(lib, t, g[0], g[1])
                e1.record(); torch.cuda.synchronize()
                times[(t, g)].append(e0.elapsed_time(e1) * 1e3 / reps)
    # builds with -DW4_CLK=1 report the clock the chip held (median over blocks of the last launch of each geometry)
    for t, lib in libs:
        if not hasattr(lib, "lfsr_w4_clk_read"): continue
        lib.lfsr_w4_clk_read.restype = c_i; lib.lfsr_w4_clk_read.argtypes = [c_p, c_i]
        for g in geoms:
            for _ in range(30): run(lib, t, g[0], g[1])
            torch.cuda.synchronize()
            buf = (C.c_ulonglong * 512)()
            assert lib.lfsr_w4_clk_read(buf, 256) == 0
            cyc = sorted(buf[2 * i] for i in range(256)); rt = sorted(buf[2 * i + 1] for i in range(256))
            print(f"clock {t:10s} n_img={g[0]} res={'y' if g[1] else 'n'}: median block {cyc[128]} shader cycles in {rt[128] * 10} ns -> {cyc[128] / (rt[128] * 10):.3f} GHz", flush=True)
    print("tag        " + "  ".join(f"{n}:{'y' if res else 'n'} med (min..max) us".rjust(30) for n, res in geoms))
    for t, _ in libs:
        cells = []
        for g in geoms:
            v = times[(t, g)]
            cells.append(f"{statistics.median(v):7.1f} ({min(v):6.1f}..{max(v):6.1f})".rjust(30))
        print(f"{t:10s} " + "  ".join(cells), flush=True)


if __name__ == "__main__":
    main()
