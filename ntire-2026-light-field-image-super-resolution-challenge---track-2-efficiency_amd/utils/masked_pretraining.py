"""Masked angular pre-training on the device: the reference's ``MaskedAngularPretraining`` /
``ProgressiveMasking`` (utils/masked_pretraining.py:34-226), same constructor arguments, strategies, 50 % skip, never-mask-the-
centre rule and ``(x_masked, mask_info)`` return; the view fill runs as one kernel through the C ABI instead of a Python loop of
slice assignments.  View selection stays on the host with Python's ``random`` exactly as upstream (``plan()`` consumes the generator
in upstream's order, so a seeded run picks the same views -- pinned on the reference in tests/golden/aux.json).
``mask_value``: 'zero' and 'mean' (each masked view filled with its own mean, upstream :121-123) run through the C ABI;
'noise' (upstream :124-127: ``torch.randn_like(view) * 0.1`` per view from torch's global generator) stays the same torch call per
view, so a seeded run draws what upstream draws on the same device."""
import random

import torch
import torch.nn as nn

from lfsr_amd import capi


class MaskedAngularPretraining(nn.Module):
    def __init__(self, angRes=5, mask_ratio=0.3, mask_strategy='random', mask_value='zero', enable_in_eval=False):
        super().__init__()
        self.angRes = angRes
        self.mask_ratio = mask_ratio
        self.mask_strategy = mask_strategy
        self.mask_value = mask_value
        self.enable_in_eval = enable_in_eval
        self.total_views = angRes * angRes
        self.num_masked = max(1, int(self.total_views * mask_ratio))
        self.center = (angRes // 2, angRes // 2)

    def _get_mask_indices(self):
        # masked_pretraining.py:140-170
        A = self.angRes
        views = [(i, j) for i in range(A) for j in range(A) if (i, j) != self.center]
        if self.mask_strategy == 'grid':
            return [(i, j) for i, j in views if (i + j) % 2 == 0][:self.num_masked]
        if self.mask_strategy == 'corners':
            corners = [(0, 0), (0, A - 1), (A - 1, 0), (A - 1, A - 1)]
            return [c for c in corners if c in views][:self.num_masked]
        if self.mask_strategy == 'center':
            d = sorted(((abs(i - self.center[0]) + abs(j - self.center[1]), (i, j)) for i, j in views), key=lambda t: t[0])
            return [v for _, v in d[:self.num_masked]]
        return random.sample(views, min(self.num_masked, len(views)))

    def plan(self):
        """Host-side decision of one call, consuming ``random`` exactly as upstream's forward does (:95-105): None = pass through."""
        if not self.training and not self.enable_in_eval:
            return None
        if random.random() > 0.5:
            return None
        return self._get_mask_indices()

    def forward(self, x):
        idx = self.plan()
        if idx is None:
            return x, {'masked': False, 'mask_ratio': 0.0}
        B, C, H, W = x.shape
        A = self.angRes
        h, w = H // A, W // A
        info = {'masked': True, 'mask_ratio': len(idx) / self.total_views, 'mask_indices': idx, 'strategy': self.mask_strategy}
        if self.mask_value == 'noise':
            out = x.clone()
            for (i, j) in idx:
                out[:, :, i * h:(i + 1) * h, j * w:(j + 1) * w] = torch.randn_like(x[:, :, i * h:(i + 1) * h, j * w:(j + 1) * w]) * 0.1
            return out, info
        if self.mask_value not in ('zero', 'mean'):
            return x.clone(), info                      # upstream: an unknown mask_value leaves the clone untouched
        flags = torch.zeros(A * A, dtype=torch.uint8)
        for (i, j) in idx:
            flags[i * A + j] = 1
        flags = flags.to(x.device)
        xin = x.detach().float().contiguous()
        out = torch.empty_like(xin)
        lib = capi.load()
        if self.mask_value == 'mean':
            fill = xin.view(B, C, A, h, A, w).mean(dim=(0, 1, 3, 5)).reshape(A * A).contiguous()    # every view's own mean
            capi.check(lib.lfsr_mask_views_fill(capi.dev_ptr(xin), capi.dev_ptr(out), capi.dev_ptr(flags), capi.dev_ptr(fill), B, C, A, h, w,
                                                capi.stream_ptr()), "mask_views_fill")
        else:
            capi.check(lib.lfsr_mask_views(capi.dev_ptr(xin), capi.dev_ptr(out), capi.dev_ptr(flags), 0.0, B, C, A, h, w,
                                           capi.stream_ptr()), "mask_views")
        return out, info


class ProgressiveMasking(nn.Module):
    # masked_pretraining.py:173-226
    def __init__(self, angRes=5, start_ratio=0.1, end_ratio=0.4, warmup_epochs=20):
        super().__init__()
        self.angRes, self.start_ratio, self.end_ratio, self.warmup_epochs = angRes, start_ratio, end_ratio, warmup_epochs
        self.current_epoch = 0
        self.masker = MaskedAngularPretraining(angRes=angRes, mask_ratio=start_ratio)

    def set_epoch(self, epoch):
        self.current_epoch = epoch
        progress = min(1.0, epoch / self.warmup_epochs)
        ratio = self.start_ratio + progress * (self.end_ratio - self.start_ratio)
        self.masker.mask_ratio = ratio
        self.masker.num_masked = max(1, int(self.masker.total_views * ratio))

    def forward(self, x):
        return self.masker(x)
