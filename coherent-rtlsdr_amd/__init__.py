"""coherent-rtlsdr_amd -- MI355X-native coherent-alignment engine (one hot path of
mlaaks/coherent-rtlsdr: the ccoherent/cdsp DSP path) behind a C ABI.

Layout:
  csrc/       hand-written HIP kernels for gfx950 + the C ABI implementation (libcrsdr.so)
  host/       C++ mirror of the reference's class surface (cdsp / csdrdevice / ccoherent /
              cpacketize) over the C ABI, plus the synthetic block source
  binding.py  ctypes view of the C ABI for tests and bench.py (PyTorch is plumbing only)
  synth.py    seeded synthetic int8 IQ block source (stands in for the dongles)
  sharding.py row sharding + the torch.distributed (RCCL) gather of the receive matrix

The hyphen in the directory name follows the task's naming; import it with
importlib.import_module("coherent-rtlsdr_amd").
"""
import importlib as _importlib


def __getattr__(name):
    if name in ("binding", "synth", "sharding"):
        return _importlib.import_module(__name__ + "." + name)
    raise AttributeError(name)
