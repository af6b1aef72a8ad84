"""Importable alias for the package directory, whose mandated name
(``ntire-2026-light-field-image-super-resolution-challenge---track-2-efficiency_amd``) is not a
valid Python identifier.  ``import lfsr_amd.capi`` resolves to files inside that directory."""
import os as _os

PKG_DIR = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                        "ntire-2026-light-field-image-super-resolution-challenge---track-2-efficiency_amd")
__path__ = [PKG_DIR]
