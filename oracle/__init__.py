"""CPU oracle for the unet_convlstm_attention hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is imported by the product
package; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may use it (as the checker / the timed CPU baseline).
"""
from .cpu_ref import *  # noqa: F401,F403
from . import data_ref  # noqa: F401
