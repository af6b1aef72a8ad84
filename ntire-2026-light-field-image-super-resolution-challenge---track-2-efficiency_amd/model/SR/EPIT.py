"""EPIT plugin (drop-in for the reference's ``model/SR/EPIT.py``): ``get_model`` / ``get_loss`` / ``weights_init`` with the
reference's state_dict key names and shapes; ``forward`` runs in the gfx950 HIP library through the C ABI.
Inference only -- the reference's own ``get_loss`` indexes ``out['SR']`` on a tensor (EPIT.py:178), i.e. EPIT training is
broken upstream, and configs name EPIT for inference."""
import math

import torch
import torch.nn as nn

from lfsr_amd import capi


class _Holder(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter container: the HIP path computes this layer")


def _conv133(cin, cout):
    return nn.Conv3d(cin, cout, kernel_size=(1, 3, 3), padding=(0, 1, 1), bias=False)


class _BasicTrans(_Holder):
    # BasicTrans.__init__, EPIT.py:75-91 (creation order kept so seeded default init matches)
    def __init__(self, channels, spa_dim, num_heads=8, dropout=0.):
        super().__init__()
        self.linear_in = nn.Linear(channels, spa_dim, bias=False)
        self.norm = nn.LayerNorm(spa_dim)
        self.attention = nn.MultiheadAttention(spa_dim, num_heads, dropout, bias=False)
        nn.init.kaiming_uniform_(self.attention.in_proj_weight, a=math.sqrt(5))
        self.attention.out_proj.bias = None
        self.attention.in_proj_bias = None
        self.feed_forward = nn.Sequential(nn.LayerNorm(spa_dim), nn.Linear(spa_dim, spa_dim * 2, bias=False), nn.ReLU(True), nn.Dropout(dropout),
                                          nn.Linear(spa_dim * 2, spa_dim, bias=False), nn.Dropout(dropout))
        self.linear_out = nn.Linear(spa_dim, channels, bias=False)


class _AltFilter(_Holder):
    # AltFilter.__init__, EPIT.py:131-142
    def __init__(self, angRes, channels):
        super().__init__()
        self.epi_trans = _BasicTrans(channels, channels * 2)
        self.conv = nn.Sequential(_conv133(channels, channels), nn.LeakyReLU(0.2, inplace=True), _conv133(channels, channels),
                                  nn.LeakyReLU(0.2, inplace=True), _conv133(channels, channels))


class get_model(nn.Module):
    def __init__(self, args):
        super().__init__()
        channels = 64
        self.angRes = args.angRes_in
        self.scale = args.scale_factor
        self.conv_init0 = nn.Sequential(_conv133(1, channels))
        self.conv_init = nn.Sequential(_conv133(channels, channels), nn.LeakyReLU(0.2, inplace=True), _conv133(channels, channels),
                                       nn.LeakyReLU(0.2, inplace=True), _conv133(channels, channels), nn.LeakyReLU(0.2, inplace=True))
        self.altblock = nn.Sequential(*[_AltFilter(self.angRes, channels) for _ in range(5)])
        self.upsampling = nn.Sequential(nn.Conv2d(channels, channels * self.scale ** 2, kernel_size=1, padding=0, bias=False),
                                        nn.PixelShuffle(self.scale), nn.LeakyReLU(0.2), nn.Conv2d(channels, 1, kernel_size=3, padding=1, bias=False))
        self._rt = None
        self._rt_version = None

    def _runtime(self, device):
        if self._rt is None:
            self._rt = capi.ModelRuntime("epit", self.angRes, self.scale, 5, 64)
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
            raise capi.LfsrError("EPIT: input must live on the MI355X (no CPU fallback in the HIP path)")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise NotImplementedError("EPIT is inference-only here (its training is broken upstream, EPIT.py:178): call under torch.no_grad()")
        return self._runtime(lr.device).forward(lr.float() if lr.dtype != torch.float32 else lr)


class get_loss(nn.Module):
    # EPIT.py:172-180 -- kept verbatim in behaviour, including the out['SR'] indexing
    def __init__(self, args):
        super().__init__()
        self.criterion_Loss = torch.nn.L1Loss()

    def forward(self, out, HR, degrade_info=None):
        return self.criterion_Loss(out['SR'], HR)


def weights_init(m):
    pass
