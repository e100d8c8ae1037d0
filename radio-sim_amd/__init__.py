"""radio-sim_amd -- MI355X (gfx950) engine for radio-sim's per-packet propagation /
delivery-verdict pass, behind the reference's RadioMedium plug-in contract.

  csrc/      HIP kernels + the C ABI (include/radiomedium_hip.h) -> libradiomedium_hip.so
  _lib.py    ctypes binding of the C ABI (fails loudly when the library or a GPU is missing)
  engine.py  thin object wrappers over one rm_context (Engine) and over a group of contexts, one per device (Group)
  host/radiomedium.hpp  C++ host-side mirror of the reference's RadioMedium / Simulator / Node API
  workload.py  synthetic inputs of SURVEY.md section 8d
  trace.py   pcap (the reference's dialect) and compact replay traces; ticks of a trace as rm_tx_record arrays
  dist.py    receiver-sharded multi-GPU tick (torch.distributed all-gather of Tx records)

There is no CPU fallback anywhere in this package.
"""
from . import _lib  # noqa: F401
from ._lib import (MODEL_NULL, MODEL_UDGM, MODEL_UDGM_CONST, MODEL_N2N, MODEL_LOGDIST,  # noqa: F401
                   UNHEARD, INTERFERED, DELIVERED, LD_SINR, MAX_BATCH, RadioMediumError, ModelParams, TxRecord,
                   TX_RECORD_DTYPE, build_library, library_path)
from .engine import Engine, Group  # noqa: F401
