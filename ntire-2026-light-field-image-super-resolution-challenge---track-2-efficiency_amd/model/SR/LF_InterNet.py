"""LF_InterNet plugin (drop-in for the reference's ``model/SR/LF_InterNet.py``): ``get_model`` / ``get_loss`` /
``weights_init`` with the reference's state_dict key names and shapes; ``forward`` runs in the gfx950 HIP
library through the C ABI (inference)."""
import torch
import torch.nn as nn

from lfsr_amd import capi


class _Holder(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter container: the HIP path computes this layer")


class _Chain(_Holder):
    # make_chains.__init__, LF_InterNet.py:45-57
    def __init__(self, A, ch):
        super().__init__()
        self.Spa2Ang = nn.Conv2d(ch, ch, kernel_size=A, stride=A, padding=0, bias=False)
        self.Ang2Spa = nn.Sequential(nn.Conv2d(ch, A * A * ch, kernel_size=1, stride=1, padding=0, bias=False), nn.PixelShuffle(A))
        self.AngConvSq = nn.Conv2d(2 * ch, ch, kernel_size=1, stride=1, padding=0, bias=False)
        self.SpaConvSq = nn.Conv2d(2 * ch, ch, kernel_size=3, stride=1, dilation=A, padding=A, bias=False)
        self.ReLU = nn.ReLU(inplace=True)


class _InterBlock(_Holder):
    def __init__(self, A, n_layers, ch):
        super().__init__()
        self.chained_layers = nn.Sequential(*[_Chain(A, ch) for _ in range(n_layers)])


class _Cascade(_Holder):
    def __init__(self, A, n_blocks, n_layers, ch):
        super().__init__()
        self.body = nn.Sequential(*[_InterBlock(A, n_layers, ch) for _ in range(n_blocks)])


class _BottleNeck(_Holder):
    # LF_InterNet.py:108-117
    def __init__(self, A, n_blocks, ch):
        super().__init__()
        self.AngBottle = nn.Conv2d(n_blocks * ch, ch, kernel_size=1, stride=1, padding=0, bias=False)
        self.Ang2Spa = nn.Sequential(nn.Conv2d(ch, A * A * ch, kernel_size=1, stride=1, padding=0, bias=False), nn.PixelShuffle(A))
        self.SpaBottle = nn.Conv2d((n_blocks + 1) * ch, ch, kernel_size=3, stride=1, dilation=A, padding=A, bias=False)
        self.ReLU = nn.ReLU(inplace=True)


class _Recon(_Holder):
    # LF_InterNet.py:128-134
    def __init__(self, A, ch, s):
        super().__init__()
        self.PreConv = nn.Conv2d(ch, ch * s ** 2, kernel_size=3, stride=1, dilation=A, padding=A, bias=False)
        self.PixelShuffle = nn.PixelShuffle(s)
        self.FinalConv = nn.Conv2d(ch, 1, kernel_size=1, stride=1, padding=0, bias=False)


class get_model(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.angRes = args.angRes_in
        channels = 64
        self.factor = args.scale_factor
        n_groups, n_blocks = 4, 4
        A = int(self.angRes)
        self.AngFE = nn.Sequential(nn.Conv2d(1, channels, kernel_size=A, stride=A, padding=0, bias=False))
        self.SpaFE = nn.Sequential(nn.Conv2d(1, channels, kernel_size=3, stride=1, dilation=A, padding=A, bias=False))
        self.CascadeInterBlock = _Cascade(A, n_groups, n_blocks, channels)
        self.BottleNeck = _BottleNeck(A, n_blocks, channels)
        self.ReconBlock = _Recon(A, channels, self.factor)
        self._rt = None
        self._rt_version = None

    def _runtime(self, device):
        if self._rt is None:
            self._rt = capi.ModelRuntime("internet", self.angRes, self.factor, 4, 4)
        ver = (device, tuple((p.data_ptr(), p._version) for p in self.parameters()))
        if ver != self._rt_version:
            self._rt.load_state(self.state_dict().items(), device)
            self._rt_version = ver
        return self._rt

    def invalidate_packed(self):
        """Force a repack at the next forward (for weight writes that bypass p._version: ``p.data.copy_``, collectives)."""
        self._rt_version = None

    def forward(self, x, Lr_info=None):
        if not x.is_cuda:
            raise capi.LfsrError("LF_InterNet: input must live on the MI355X (no CPU fallback in the HIP path)")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise NotImplementedError("LF_InterNet HIP backward is not built: call under torch.no_grad()")
        return self._runtime(x.device).forward(x.float() if x.dtype != torch.float32 else x)


def weights_init(m):
    pass


class get_loss(nn.Module):
    # LF_InterNet.py:176-186
    def __init__(self, args):
        super().__init__()
        self.criterion_Loss = torch.nn.L1Loss()

    def forward(self, SR, HR, criterion_data=[]):
        return self.criterion_Loss(SR, HR)
