"""LFT plugin (drop-in for the reference's ``model/SR/LFT.py``): ``get_model`` / ``get_loss`` / ``weights_init`` with the
reference's state_dict key names and shapes; ``forward`` runs in the gfx950 HIP library through the C ABI
(inference; the backward of the transformer models is not built)."""
import math

import torch
import torch.nn as nn

from lfsr_amd import capi


class _Holder(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter container: the HIP path computes this layer")


def _conv133(cin, cout):
    return nn.Conv3d(cin, cout, kernel_size=(1, 3, 3), padding=(0, 1, 1), dilation=1, bias=False)


def _ffn(dim, dropout):
    return nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, dim * 2, bias=False), nn.ReLU(True), nn.Dropout(dropout),
                         nn.Linear(dim * 2, dim, bias=False), nn.Dropout(dropout))


class _SpaTrans(_Holder):
    # SpaTrans.__init__, LFT.py:134-158
    def __init__(self, channels, heads, dropout):
        super().__init__()
        spa_dim = channels * 2
        self.MLP = nn.Linear(channels * 9, spa_dim, bias=False)
        self.norm = nn.LayerNorm(spa_dim)
        self.attention = nn.MultiheadAttention(spa_dim, heads, dropout, bias=False)
        nn.init.kaiming_uniform_(self.attention.in_proj_weight, a=math.sqrt(5))
        self.attention.out_proj.bias = None
        self.attention.in_proj_bias = None
        self.feed_forward = _ffn(spa_dim, dropout)
        self.linear = nn.Sequential(nn.Conv3d(spa_dim, channels, kernel_size=(1, 1, 1), padding=(0, 0, 0), dilation=1, bias=False))


class _AngTrans(_Holder):
    # AngTrans.__init__, LFT.py:207-224
    def __init__(self, channels, heads, dropout):
        super().__init__()
        self.norm = nn.LayerNorm(channels)
        self.attention = nn.MultiheadAttention(channels, heads, dropout, bias=False)
        nn.init.kaiming_uniform_(self.attention.in_proj_weight, a=math.sqrt(5))
        self.attention.out_proj.bias = None
        self.feed_forward = _ffn(channels, dropout)


class _AltFilter(_Holder):
    # AltFilter.__init__, LFT.py:249-254 (spa_trans is created before ang_trans)
    def __init__(self, channels):
        super().__init__()
        self.spa_trans = _SpaTrans(channels, 8, 0.)
        self.ang_trans = _AngTrans(channels, 8, 0.)


class get_model(nn.Module):
    def __init__(self, args):
        super().__init__()
        channels = 64
        self.channels = channels
        self.angRes = args.angRes_in
        self.factor = args.scale_factor
        self.conv_init0 = nn.Sequential(_conv133(1, channels))
        self.conv_init = nn.Sequential(_conv133(channels, channels), nn.LeakyReLU(0.2, inplace=True), _conv133(channels, channels),
                                       nn.LeakyReLU(0.2, inplace=True), _conv133(channels, channels), nn.LeakyReLU(0.2, inplace=True))
        self.altblock = nn.Sequential(*[_AltFilter(channels) for _ in range(4)])
        self.upsampling = nn.Sequential(nn.Conv2d(channels, channels * self.factor ** 2, kernel_size=1, padding=0, dilation=1, bias=False),
                                        nn.PixelShuffle(self.factor), nn.LeakyReLU(0.2), nn.Conv2d(channels, 1, kernel_size=3, stride=1, padding=1, bias=False))
        self._rt = None
        self._rt_version = None

    def _runtime(self, device):
        if self._rt is None:
            self._rt = capi.ModelRuntime("lft", self.angRes, self.factor, 4, 64)
        ver = (device, tuple((p.data_ptr(), p._version) for p in self.parameters()))
        if ver != self._rt_version:
            self._rt.load_state(self.state_dict().items(), device)
            self._rt_version = ver
        return self._rt

    def invalidate_packed(self):
        """Force a repack at the next forward (for weight writes that bypass p._version: ``p.data.copy_``, collectives)."""
        self._rt_version = None

    def forward(self, lr, info=None):
        if not lr.is_cuda:
            raise capi.LfsrError("LFT: input must live on the MI355X (no CPU fallback in the HIP path)")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise NotImplementedError("LFT HIP backward is not built: call under torch.no_grad()")
        return self._runtime(lr.device).forward(lr.float() if lr.dtype != torch.float32 else lr)


class get_loss(nn.Module):
    # LFT.py:276-286
    def __init__(self, args):
        super().__init__()
        self.criterion_Loss = torch.nn.L1Loss()

    def forward(self, SR, HR, info=None):
        return self.criterion_Loss(SR, HR)


def weights_init(m):
    pass
