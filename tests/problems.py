"""Problem definitions shared by the parity tests: the workload builders of the product package (mpp_amd.workloads)
plus the adapters that build the same problems on the CPU oracle (oracle/adapters.py)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from mpp_amd.workloads import *  # noqa: E402,F401,F403
from mpp_amd.workloads import _mms_funcs, _restrict_conn_set  # noqa: E402,F401
from oracle.adapters import th_mms, th_set_conditions, to_oracle, to_th_oracle, to_thermal_oracle  # noqa: E402,F401
