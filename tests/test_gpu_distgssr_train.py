"""GPU: DistgSSR backward (d10/d11) through the C ABI.
Gates: loss equal to the reference's; every parameter gradient within rel-L2 1e-4 of the reference's
fp32 autograd (golden: per-parameter norm, random projection, and full tensors for the small ones; plus a live
comparison against autograd over the torch-CPU form of the oracle)."""
import json
import os
import sys

import numpy as np
import pytest
import torch

from lfsr_amd import capi
from lfsr_amd.synth import synth_input, synth_state_dict
from tests.helpers import GOLDEN, model_case

pytestmark = pytest.mark.gpu
torch.set_num_threads(8)   # the CPU reference legs run tiny convs: 100+ default threads on the GPU box only contend


def load_plugin():
    import importlib
    sys.path.insert(0, capi._HERE)
    try:
        return importlib.import_module("model.SR.DistgSSR")
    finally:
        sys.path.remove(capi._HERE)


def build(M, A, s, sd):
    from argparse import Namespace
    net = M.get_model(Namespace(angRes_in=A, angRes_out=A, scale_factor=s))
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return net.to("cuda:0").train()


def test_grads_vs_reference_golden():
    meta = json.load(open(os.path.join(GOLDEN, "distg_grads.json")))
    gold = np.load(os.path.join(GOLDEN, "distg_grads.npz"))
    A, h, w, s, B = meta["A"], meta["h"], meta["w"], meta["s"], meta["B"]
    case, sd, x, _ = model_case("DistgSSR", "a3h6w8s2")
    M = load_plugin()
    net = build(M, A, s, sd)
    label = torch.from_numpy(synth_input((B, 1, A * h * s, A * w * s), seed=meta["label_seed"])).cuda()
    out = net(torch.from_numpy(x).cuda(), [A, A])
    loss = M.get_loss(None)(out, label, [A, A])
    loss.backward()
    assert abs(loss.item() - float(gold["loss"])) < 1e-6
    names = [k for k, _ in net.named_parameters()]
    assert names == meta["names"]
    worst = 0.0
    for i, (k, p) in enumerate(net.named_parameters()):
        g = p.grad.detach().cpu().numpy().astype(np.float64)
        assert np.isfinite(g).all(), k
        n_ref, pr_ref = gold["norms"][i], gold["projs"][i]
        probe = np.random.default_rng([7, i]).standard_normal(g.shape)
        n, pr = np.sqrt((g * g).sum()), (g * probe).sum()
        assert abs(n - n_ref) <= 1e-4 * n_ref + 1e-12, (k, n, n_ref)
        # |<g - g_ref, probe>| <= ||g - g_ref|| ||probe||  with rel-L2 <= 1e-4
        assert abs(pr - pr_ref) <= 1e-4 * n_ref * np.sqrt((probe * probe).sum()) + 1e-12, (k, pr, pr_ref)
        if "grad::" + k in gold.files:
            ref = gold["grad::" + k].astype(np.float64)
            rel = np.sqrt(((g - ref) ** 2).sum()) / max(np.sqrt((ref ** 2).sum()), 1e-30)
            worst = max(worst, rel)
            assert rel <= 1e-4, (k, rel)
    assert worst > 0.0   # at least one full-tensor comparison happened


@pytest.mark.parametrize("A,h,w,s,B", [(5, 8, 8, 4, 1), (3, 6, 8, 2, 2)])
def test_grads_vs_torch_port_autograd(A, h, w, s, B):
    """every parameter, full tensors, against autograd over stock torch CPU ops (oracle/lfsr_torch_port.py)"""
    from oracle import lfsr_torch_port as T
    tag = "a5h8s4" if A == 5 else "a3h6w8s2"
    case, sd, x, _ = model_case("DistgSSR", tag)
    M = load_plugin()
    net = build(M, A, s, sd)
    label_np = synth_input((B, 1, A * h * s, A * w * s), seed=2)
    out = net(torch.from_numpy(x).cuda(), None)
    loss = torch.nn.functional.l1_loss(out, torch.from_numpy(label_np).cuda())
    loss.backward()
    sdt = {k: torch.from_numpy(v).double().requires_grad_(True) for k, v in sd.items()}
    with torch.enable_grad():
        ref_out = T.distgssr_forward_graph(torch.from_numpy(x).double(), sdt, A, s)
        ref_loss = torch.nn.functional.l1_loss(ref_out, torch.from_numpy(label_np).double())
    ref_loss.backward()
    assert abs(loss.item() - ref_loss.item()) < 1e-6
    rels = {}
    for k, p in net.named_parameters():
        g = p.grad.detach().cpu().double()
        r = sdt[k].grad
        rels[k] = float((g - r).norm() / r.norm().clamp_min(1e-30))
    v = np.array(sorted(rels.values()))
    print("rel-L2 grad error vs fp64 autograd: median %.2e  p90 %.2e  max %.2e (%s)" % (
        np.median(v), v[int(0.9 * len(v))], v[-1], max(rels, key=rels.get)))
    # the 1e-4 gate is held against the reference's own fp32 gradients (test above); against an fp64 oracle an
    # isolated LeakyReLU' flip at a pre-activation within fp32 round-off of zero moves single small tensors by up to a few
    # 1e-3 (the Winograd-form 3x3 convs carry ~2x the round-off of the direct form, hence ~2x the flips)
    assert np.median(v) <= 5e-5 and v[-1] <= 1e-2, rels
    # the flat bucket holds the same numbers in state_dict order (what the RCCL all-reduce sees)
    flat = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
    assert torch.equal(flat, net.grad_bucket)


def test_train_step_matches_inference_forward():
    case, sd, x, npz = model_case("DistgSSR", "a3h6w8s2")
    M = load_plugin()
    net = build(M, 3, 2, sd)
    y_train = net(torch.from_numpy(x).cuda(), None)
    with torch.no_grad():
        y_inf = net(torch.from_numpy(x).cuda(), None)
    assert np.abs(y_train.detach().cpu().numpy() - npz["a3h6w8s2_out"]).max() < 1e-4
    assert torch.allclose(y_train.detach(), y_inf, atol=1e-6)
    # optimizer step changes parameter versions -> weights are repacked
    opt = torch.optim.AdamW(net.parameters(), lr=1e-3)
    y_train.abs().mean().backward()
    opt.step()
    with torch.no_grad():
        y2 = net(torch.from_numpy(x).cuda(), None)
    assert not torch.allclose(y2, y_inf, atol=1e-6)


def test_full_train_step_vs_cpu_reference():
    """train.py:243-268 in fp32: L1 -> backward -> clip_grad_norm_(1.0) -> AdamW(lr 2e-4, wd 1e-4); parameters after
    two steps equal those of the same loop over the torch-CPU form of the oracle."""
    from lfsr_amd.train_step import train_step
    from oracle import lfsr_torch_port as T
    A, h, w, s, B = 3, 6, 8, 2, 2
    case, sd, x, _ = model_case("DistgSSR", "a3h6w8s2")
    label = synth_input((B, 1, A * h * s, A * w * s), seed=2)
    M = load_plugin()
    net = build(M, A, s, sd)
    crit = M.get_loss(None)
    opt = torch.optim.AdamW(net.parameters(), lr=2e-4, weight_decay=1e-4)
    ref = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in sd.items()}
    ropt = torch.optim.AdamW(list(ref.values()), lr=2e-4, weight_decay=1e-4)
    losses, rlosses = [], []
    for step in range(2):
        loss, _ = train_step(net, crit, opt, torch.from_numpy(x).cuda(), torch.from_numpy(label).cuda())
        losses.append(loss.item())
        ropt.zero_grad()
        with torch.enable_grad():
            rl = torch.nn.functional.l1_loss(T.distgssr_forward_graph(torch.from_numpy(x), ref, A, s), torch.from_numpy(label))
        rl.backward()
        torch.nn.utils.clip_grad_norm_(list(ref.values()), 1.0)
        ropt.step()
        rlosses.append(rl.item())
    assert np.allclose(losses, rlosses, atol=1e-6)
    # AdamW's first updates are ~ lr * sign(g): an element whose gradient is at round-off level may move by +-lr in
    # either run, so compare robustly: (almost) all elements to 2e-6, none further than the two steps can take it
    bad = tot = 0
    for k, p in net.named_parameters():
        d = (p.detach().cpu() - ref[k].detach()).abs()
        assert float(d.max()) <= 2 * 2 * 2e-4 + 1e-6, k
        bad += int((d > 2e-6).sum())
        tot += d.numel()
    assert bad / tot < 1e-3, (bad, tot)


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE geometry (configs[3]: 5x5 views of 32x32, x4): the code paths the toy geometries never reach -- persistent weight-gradient
# blocks accumulating over several tiles, the masked Winograd data gradient over hundreds of tiles, >= 8-way split reductions
# ---------------------------------------------------------------------------------------------------------------------

def _full_net():
    case, sd, _, _ = model_case("DistgSSR", "full")
    M = load_plugin()
    return M, build(M, 5, 4, sd), sd


def test_grads_full_geometry_vs_torch_port_autograd():
    """(A,h,w,s,B) = (5,32,32,4,2): every one of the 137 gradients at the BASELINE geometry (reference: train.py:256-264).
    Truth = fp64 autograd over the stock-torch CPU form of the oracle; yardstick = the SAME graph in fp32 (what the reference computes on the CPU).

    What round 3 measured (tools/grad_parity.py -> profiles/r03_grad_parity.json): at this size NEITHER implementation holds rel-L2 <= 1e-4 on every
    tensor against fp64 (reference fp32: max 3-4e-4; HIP in every kernel selection, all-direct included: max 3-4e-4), and the cause is not arithmetic:
    a few dozen of the 234 M LeakyReLU decisions sit within round-off of zero and come out on the other side (reference fp32: ~30 flips, HIP 33-71
    depending on the conv form), and ONE flipped decision moves a small angular / epipolar weight tensor by 1-3e-4.  Which tensors are hit is chance,
    so "per tensor <= 3 x the reference's error on that tensor" is ill-posed (the reference's own e_ref ranges over 2e-6 ... 4e-4 by where ITS flips
    fell).  The error therefore is gated in its two parts, each against the reference's own figure for that part:
      (i)  arithmetic: against the fp64 graph evaluated with the HIP forward's own LeakyReLU decisions (oracle `force=`), every tensor holds SURVEY's
           rel-L2 <= 1e-4 (or 3 x the reference-fp32 figure of the same construction, which matters for upsample.0.bias only: a sum of +-1/N
           that cancels to 1e-4 of its terms);
      (ii) decisions: the HIP forward flips at most 4 x as many masks as the reference's fp32 forward does (and < 1e-6 of all decisions);
      (iii) the total error's distribution stays within 8 x / 3 x / 3 x of the reference's median / p90 / max (measured: 2.4-3.7 x / 1.3 x / 1.0-1.4 x --
            the median follows the flip count, 71 against 30, and the reference's own median moves between 4.4e-6 and 6.8e-6 with its thread count).
    Measured (profiles/r03_grad_parity.json): HIP arithmetic-only error max 6.7e-7 over all 137 tensors (median 1.6e-7; the reference's fp32: 1.7e-6, and
    1.5e-4 on upsample.0.bias), in every kernel selection; everything above that is the flipped decisions."""
    from oracle import lfsr_torch_port as T
    from tests import helpers as TH
    A, h, w, s, B = 5, 32, 32, 4, 2
    M, net, sd = _full_net()
    x = synth_input((B, 1, A * h, A * w), seed=1)
    label = synth_input((B, 1, A * h * s, A * w * s), seed=2)
    xa = torch.from_numpy(x).cuda()
    out = net(xa, None)
    loss = M.get_loss(None)(out, torch.from_numpy(label).cuda(), None)
    blocks = [f"disentg.Group.{g}.Block.{b}." for g in range(4) for b in range(4)]
    hip_masks = {pre + k: TH.hip_saved_mask(net._rt, xa, k, i) for i, pre in enumerate(blocks) for k in TH.MASK_KINDS}    # flat, HIP order; before backward
    loss.backward()
    g_hip = {k: p.grad.detach().cpu().double() for k, p in net.named_parameters()}
    assert all(torch.isfinite(g).all() for g in g_hip.values())

    def cpu_graph(dt, force=None):
        sdt = {k: torch.from_numpy(v).to(dt).requires_grad_(True) for k, v in sd.items()}
        rec = {} if force is None else None
        with torch.enable_grad():
            rl = torch.nn.functional.l1_loss(T.distgssr_forward_graph(torch.from_numpy(x).to(dt), sdt, A, s, rec=rec, force=force), torch.from_numpy(label).to(dt))
        rl.backward()
        assert abs(loss.item() - rl.item()) < 1e-6
        return {k: v.grad.double() for k, v in sdt.items()}, rec
    g64, rec64 = cpu_graph(torch.float64)
    g32, rec32 = cpu_graph(torch.float32)
    nrm = {k: g.norm().clamp_min(1e-30) for k, g in g64.items()}
    rel = lambda a, b: {k: float((a[k] - b[k]).norm() / nrm[k]) for k in nrm}
    # (ii) decisions
    n_dec = sum(m.numel() for m in hip_masks.values())
    flips_hip = sum(int((hip_masks[k] != TH.ref_mask_to_hip(rec64[k], k.rsplit(".", 1)[1], B, A, h, w)).sum()) for k in hip_masks)
    flips_ref = sum(int((rec32[k] != rec64[k]).sum()) for k in rec64)
    # (i) arithmetic: the same graph with each implementation's own decisions
    g64_hipmask, _ = cpu_graph(torch.float64, force={k: TH.hip_mask_to_ref(m, k.rsplit(".", 1)[1], B, A, h, w) for k, m in hip_masks.items()})
    g64_refmask, _ = cpu_graph(torch.float64, force=rec32)
    del rec64, rec32, hip_masks
    ea_hip, ea_ref = rel(g_hip, g64_hipmask), rel(g32, g64_refmask)
    e_hip, e_ref = rel(g_hip, g64), rel(g32, g64)
    eh, er = np.array(sorted(e_hip.values())), np.array(sorted(e_ref.values()))
    p90 = lambda v: v[int(0.9 * len(v))]
    print("full geometry, rel-L2 vs fp64 autograd: HIP median %.2e p90 %.2e max %.2e | reference fp32 CPU median %.2e p90 %.2e max %.2e" % (
        np.median(eh), p90(eh), eh[-1], np.median(er), p90(er), er[-1]))
    print("LeakyReLU decisions flipped vs fp64: HIP %d, reference fp32 %d of %d" % (flips_hip, flips_ref, n_dec))
    print("arithmetic-only error (own decisions forced into the fp64 graph): HIP max %.2e (%s) median %.2e | reference fp32 max %.2e median %.2e" % (
        max(ea_hip.values()), max(ea_hip, key=ea_hip.get), np.median(list(ea_hip.values())), max(ea_ref.values()), np.median(list(ea_ref.values()))))
    bad = {k: (ea_hip[k], ea_ref[k]) for k in ea_hip if ea_hip[k] > max(1e-4, 3.0 * ea_ref[k])}
    assert not bad, bad                                                                   # (i)
    assert flips_hip <= 4 * max(flips_ref, 16) and flips_hip < 1e-6 * n_dec               # (ii)
    assert np.median(eh) <= 8 * np.median(er) and p90(eh) <= 3 * p90(er) and eh[-1] <= 3 * er[-1]   # (iii)


def test_grad_bucket_b8_is_mean_of_b1_buckets():
    """configs[3]'s per-GPU batch: the B = 8 bucket equals the mean of the eight B = 1 buckets (L1 'mean' loss; linearity of the
    backward in the batch) to fp32 round-off -- a size-independent property at the bench geometry."""
    A, h, w, s, B = 5, 32, 32, 4, 8
    M, net, _ = _full_net()
    crit = M.get_loss(None)
    x = torch.from_numpy(synth_input((B, 1, A * h, A * w), seed=11)).cuda()
    y = torch.from_numpy(synth_input((B, 1, A * h * s, A * w * s), seed=12)).cuda()
    crit(net(x, None), y, None).backward()
    big = net.grad_bucket.double().clone()
    acc = torch.zeros_like(big)
    for i in range(B):
        net.zero_grad(set_to_none=True)
        crit(net(x[i:i + 1], None), y[i:i + 1], None).backward()
        acc += net.grad_bucket.double()
    acc /= B
    spans = net._spans
    worst = max(float((big[o:o + n] - acc[o:o + n]).norm() / acc[o:o + n].norm().clamp_min(1e-30)) for o, n in spans.values())
    print("B=8 bucket vs mean of 8 B=1 buckets: worst per-parameter rel-L2 %.2e" % worst)
    assert worst <= 2e-5


def test_accumulation_and_zero_grad_in_place():
    """Gradient accumulation over two micro-batches and zero_grad(set_to_none=False) (p.grad kept and zeroed in place): neither may
    alias a buffer the next backward overwrites."""
    from oracle import lfsr_torch_port as T
    A, h, w, s = 3, 6, 8, 2
    case, sd, x, _ = model_case("DistgSSR", "a3h6w8s2")
    M = load_plugin()
    net = build(M, A, s, sd)
    xs = [torch.from_numpy(x[i:i + 1]) for i in range(2)]
    ys = [torch.from_numpy(synth_input((1, 1, A * h * s, A * w * s), seed=20 + i)) for i in range(2)]
    ref = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in sd.items()}
    for xi, yi in zip(xs, ys):     # accumulate: g0 + g1
        torch.nn.functional.l1_loss(net(xi.cuda(), None), yi.cuda()).backward()
        with torch.enable_grad():
            torch.nn.functional.l1_loss(T.distgssr_forward_graph(xi, ref, A, s), yi).backward()
    for k, p in net.named_parameters():
        r = ref[k].grad
        assert float((p.grad.cpu() - r).norm() / r.norm().clamp_min(1e-30)) <= 1e-4, k
    net.zero_grad(set_to_none=False)   # p.grad zeroed in place, same tensors
    for r in ref.values():
        r.grad.zero_()
    torch.nn.functional.l1_loss(net(xs[1].cuda(), None), ys[1].cuda()).backward()
    with torch.enable_grad():
        torch.nn.functional.l1_loss(T.distgssr_forward_graph(xs[1], ref, A, s), ys[1]).backward()
    for k, p in net.named_parameters():
        r = ref[k].grad
        assert float((p.grad.cpu() - r).norm() / r.norm().clamp_min(1e-30)) <= 1e-4, (k, "after zero_grad(set_to_none=False)")


def test_second_forward_before_backward_is_rejected():
    case, sd, x, _ = model_case("DistgSSR", "a3h6w8s2")
    M = load_plugin()
    net = build(M, 3, 2, sd)
    xa = torch.from_numpy(x).cuda()
    o1 = net(xa, None)
    o2 = net(xa * 0.5, None)          # overwrites the one training workspace
    with pytest.raises(capi.LfsrError):
        o1.abs().mean().backward()
    o2.abs().mean().backward()        # the latest graph is fine
    assert all(torch.isfinite(p.grad).all() for p in net.parameters())


def test_amp_loop_shape_gradscaler():
    """The reference's unchanged AMP loop shape (train.py:127,256-268): autocast + GradScaler.scale / unscale_ / clip / step around the
    HIP autograd node.  The HIP path computes in fp32 whatever the autocast dtype; with an fp32 loss the scaler's scale cancels exactly,
    so two steps equal two steps of the plain fp32 loop."""
    A, h, w, s, B = 3, 6, 8, 2, 2
    case, sd, x, _ = model_case("DistgSSR", "a3h6w8s2")
    M = load_plugin()
    label = torch.from_numpy(synth_input((B, 1, A * h * s, A * w * s), seed=2)).cuda()
    xa = torch.from_numpy(x).cuda()
    nets = [build(M, A, s, sd) for _ in range(2)]
    opts = [torch.optim.AdamW(n.parameters(), lr=2e-4, weight_decay=1e-4) for n in nets]
    crit = M.get_loss(None)
    scaler = torch.amp.GradScaler("cuda", init_scale=1024.0)
    for _ in range(2):
        opts[0].zero_grad()
        with torch.autocast("cuda", dtype=torch.float16):
            out = nets[0](xa, None)
            loss = crit(out, label, None)
        assert out.dtype == torch.float32
        scaler.scale(loss).backward()
        scaler.unscale_(opts[0])
        torch.nn.utils.clip_grad_norm_(nets[0].parameters(), max_norm=1.0)
        scaler.step(opts[0])
        scaler.update()
        opts[1].zero_grad()
        l1 = crit(nets[1](xa, None), label, None)
        l1.backward()
        torch.nn.utils.clip_grad_norm_(nets[1].parameters(), max_norm=1.0)
        opts[1].step()
        assert abs(loss.item() - l1.item()) < 1e-6
    for (k, p), (_, q) in zip(nets[0].named_parameters(), nets[1].named_parameters()):
        assert float((p - q).abs().max()) <= 2 * 2e-4 + 1e-6, k      # AdamW's first updates ~ lr * sign(g): round-off level gradients may differ in sign
    diff = sum(int(((p - q).abs() > 2e-6).sum()) for p, q in zip(nets[0].parameters(), nets[1].parameters()))
    assert diff / sum(p.numel() for p in nets[0].parameters()) < 1e-3


def test_repack_modes_equal_eager(monkeypatch):
    """After an optimizer step only parameter VALUES change (model/SR/DistgSSR.py:_repack): the weight repack runs as ONE launch per pack kind from a
    device-side descriptor table (default), or is replayed from a captured graph (LFSR_PACK_BATCH=0), or is re-issued launch by launch
    (LFSR_PACK_BATCH=0 LFSR_PACK_GRAPH=0).  Five training steps in each mode are equal bit for bit -- for angRes 3 (generic 3x3 packs) and angRes 5
    (the F(2,5) EPI pack as well)."""
    from lfsr_amd.train_step import train_step
    for tag, A, s in (("a3h6w8s2", 3, 2), ("a5h8s4", 5, 4)):
        case, sd, x, _ = model_case("DistgSSR", tag)
        xa = torch.from_numpy(x).cuda()
        label = torch.from_numpy(synth_input((xa.shape[0], 1, xa.shape[2] * s, xa.shape[3] * s), seed=2)).cuda()
        M = load_plugin()
        runs = []
        for batch, graph in (("1", "1"), ("0", "1"), ("0", "0")):
            monkeypatch.setenv("LFSR_PACK_BATCH", batch)
            monkeypatch.setenv("LFSR_PACK_GRAPH", graph)
            net = build(M, A, s, sd)
            opt = torch.optim.AdamW(net.parameters(), lr=2e-4, weight_decay=1e-4)
            crit = M.get_loss(None)
            losses = [float(train_step(net, crit, opt, xa, label)[0]) for _ in range(5)]
            assert (net._pack_graph is not None) == (batch == "0" and graph == "1")
            runs.append((losses, [p.detach().clone() for p in net.parameters()]))
        for r in runs[1:]:
            assert runs[0][0] == r[0], tag
            assert all(torch.equal(a, b) for a, b in zip(runs[0][1], r[1])), tag
        assert runs[0][0][-1] < runs[0][0][0]      # and it trains
    monkeypatch.delenv("LFSR_PACK_BATCH"); monkeypatch.delenv("LFSR_PACK_GRAPH")


@pytest.mark.parametrize("ragged", [False, True])
def test_line_form_gradient_kernels_equal_the_gather_forms(monkeypatch, ragged):
    """the round-2 backward kernels (Winograd-form 3x3 weight gradient, EPI-line weight / data gradients of EPIConv.0, streaming AngConv.0 data gradient,
    row-GEMM fuse.0 data gradient) against the direct / gather-GEMM forms they replaced, selected by environment variables: the same gradients to fp32
    round-off for every parameter (angRes 5, the ragged-free reduced geometry and a batch of 2)"""
    case, sd, x, _ = model_case("DistgSSR", "a5h8s4")
    A, s = 5, 4
    M = load_plugin()
    if ragged:     # h = 6, w = 9: EPI lines of different lengths in the two passes, odd width, a half-filled 2-row tile pair in the Winograd weight gradient
        xb = torch.from_numpy(synth_input((2, 1, A * 6, A * 9), seed=11)).cuda()
    else:
        xb = torch.from_numpy(np.concatenate([x, 0.5 * x[:, :, ::-1].copy()], 0)).cuda()
    label = torch.from_numpy(synth_input((2, 1, xb.shape[2] * s, xb.shape[3] * s), seed=4)).cuda()
    def grads(env):
        for k in ("LFSR_WGRAD3", "LFSR_WGRAD_EPI", "LFSR_DGRAD_EPI", "LFSR_DGRAD_ANG", "LFSR_WGRAD_PW", "LFSR_DGRAD_PW"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        net = build(M, A, s, sd)
        torch.nn.functional.l1_loss(net(xb, None), label).backward()
        torch.cuda.synchronize()
        return {k: p.grad.detach().double().cpu() for k, p in net.named_parameters()}
    new = grads({})
    # (backward selectors only: both runs share ONE forward, hence the same LeakyReLU decisions -- what differs is the summation order of the gradient kernels)
    old = grads({"LFSR_WGRAD3": "direct", "LFSR_WGRAD_EPI": "gather", "LFSR_DGRAD_EPI": "gather", "LFSR_DGRAD_ANG": "gather", "LFSR_WGRAD_PW": "gather", "LFSR_DGRAD_PW": "gather"})
    for k in ("LFSR_WGRAD3", "LFSR_WGRAD_EPI", "LFSR_DGRAD_EPI", "LFSR_DGRAD_ANG", "LFSR_WGRAD_PW", "LFSR_DGRAD_PW"):
        monkeypatch.delenv(k, raising=False)
    worst = 0.0
    for k in new:
        rel = float((new[k] - old[k]).norm() / old[k].norm().clamp_min(1e-30))
        worst = max(worst, rel)
        assert rel <= 2e-5, (k, rel)        # two fp32 evaluation orders of the same sums over the same forward
    assert worst > 0.0                      # the selections really ran different kernels
