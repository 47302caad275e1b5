"""Measurement tools run on the -DIB_AB build of the library (environment A/B switches, in-kernel stamp hooks):
`import tools._ab` (or `from tools import _ab`) BEFORE importing inferbiomechanics_amd.hip selects it through IB_HIP_LIB,
unless the caller already chose a library."""
import os

_HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
AB_LIB = os.path.join(_HERE, "inferbiomechanics_amd", "lib", "ab", "libib_hip_ab.so")
os.environ.setdefault("IB_HIP_LIB", AB_LIB)
