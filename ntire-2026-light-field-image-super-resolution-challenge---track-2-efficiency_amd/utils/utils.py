"""Hot-path half of the reference's ``utils/utils.py``: ``ImageExtend`` / ``LFdivide`` / ``LFintegrate``
(utils/utils.py:137-178) with the same names, argument order and meaning, running on the MI355X through the
C ABI.  The metric / logging / colour helpers of that file are host code outside this path (SURVEY section 2
rows 8, 11) and are not mirrored; importing this module does NOT parse ``sys.argv`` (the reference's does,
utils/utils.py:8 -> option.py:36)."""
from lfsr_amd import capi


def ImageExtend(Im, bdr):
    return capi.image_extend(Im.contiguous(), list(bdr))


def LFdivide(data, angRes, patch_size, stride):
    return capi.lf_divide(data.contiguous(), angRes, patch_size, stride)


def LFintegrate(subLF, angRes, pz, stride, h, w):
    return capi.lf_integrate(subLF.contiguous(), angRes, pz, stride, h, w)
