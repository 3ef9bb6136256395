"""GPU helper: print the protocol-timeout counter of the library selected by FINCFLOW_LIB."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fincflow_amd import _lib
print("timeouts", _lib.hlp_timeouts())
