"""``cal_metrics`` of the reference's ``utils/utils.py:91-134`` on the device: same signature, same
return value ``(PSNR_mean, SSIM_mean)`` (means over the views whose value is > 0, as upstream), but label and output stay on
the MI355X: one kernel computes all B x A x A per-view PSNR / SSIM values in fp64; only B*A*A*2 doubles cross PCIe."""
import torch

from lfsr_amd import capi


def view_metrics(label, out, angRes, want_ssim=True):
    """label, out: (B,1,A*H,A*W) SAI mosaics on the device -> (psnr, ssim) tensors of shape (B, A, A), float64."""
    lib = capi.load()
    B, c1, Hh, Ww = label.shape
    if c1 != 1 or out.shape != label.shape:
        raise capi.LfsrError(f"cal_metrics expects two (B,1,A*H,A*W) tensors, got {tuple(label.shape)} {tuple(out.shape)}")
    H, W = Hh // angRes, Ww // angRes
    label = label.detach().float().contiguous()
    out = out.detach().float().contiguous()
    psnr = torch.empty(B * angRes * angRes, dtype=torch.float64, device=label.device)
    ssim = torch.empty_like(psnr) if want_ssim else None
    capi.check(lib.lfsr_view_metrics(capi.dev_ptr(label), capi.dev_ptr(out), capi.dev_ptr(psnr), capi.dev_ptr(ssim) if want_ssim else None,
                                     B, angRes, H, W, capi.stream_ptr()), "view_metrics")
    shape = (B, angRes, angRes)
    return psnr.view(shape), (ssim.view(shape) if want_ssim else None)


def cal_metrics(args, label, out):
    """utils/utils.py:91-134 for task 'SR' (4-D SAI-mosaic inputs, train.py:273,322)."""
    if label.dim() != 4:
        raise capi.LfsrError("cal_metrics (device): only the 4-D (B,1,A*H,A*W) form used by the SR task is built")
    psnr, ssim = view_metrics(label, out, args.angRes_in)
    psnr, ssim = psnr.cpu(), ssim.cpu()
    vp, vs = int((psnr > 0).sum()), int((ssim > 0).sum())
    return (float(psnr.sum() / vp) if vp > 0 else 0.0), (float(ssim.sum() / vs) if vs > 0 else 0.0)
