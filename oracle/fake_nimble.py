"""A closed-form stand-in for the part of `nimblephysics` the reference window loader touches.

TEST INFRASTRUCTURE ONLY (oracle/): imported by oracle/make_golden.py -- where it is installed as
``sys.modules['nimblephysics']`` under the REAL reference ``AddBiomechanicsDataset``
(src/data/AddBiomechanicsDataset.py:63-139 index, :161-285 ``__getitem__``) to produce
tests/golden/loader_windows.npz -- and by tests/, where the same module is installed under this repo's
loader so both read identical "files".  nimblephysics itself (a C++ wheel) is not installed in the image.

API surface reproduced (the calls the reference makes, nothing more):
``biomechanics.SubjectOnDisk(path)`` with ``getNumDofs getGroundForceBodies getNumTrials getTrialLength
getMissingGRF getMassKg getNumProcessingPasses readSkel readFrames``; ``biomechanics.MissingGRFReason``;
frames with ``processingPasses[i].<field>`` float64 vectors.

A "file" is an EMPTY ``<name>.b3d``; its content is the closed form below keyed by the base name, so nothing
binary is committed and both sides regenerate the same values.
"""
import enum
import os
import types
from typing import Dict, List

import numpy as np


class MissingGRFReason(enum.Enum):
    notMissingGRF = 0
    measuredGrfZeroWhenAccelerationNonZero = 1
    unmeasuredExternalForceDetected = 2


# name -> subject description.  `missing[t]` lists (first, last) inclusive frame ranges with a missing-GRF reason.
SUBJECTS: Dict[str, dict] = {
    "alpha": dict(seed=1, mass=72.4, bodies=["calcn_l", "calcn_r"], trials=[130, 61, 40],
                  missing={0: [(20, 24), (90, 90)], 1: []}, passes=2),
    "beta": dict(seed=2, mass=58.13, bodies=["pelvis", "calcn_r", "calcn_l"], trials=[75, 140],
                 missing={0: [(0, 3)], 1: [(64, 64), (137, 139)]}, passes=3),
    "gamma": dict(seed=3, mass=91.027, bodies=["calcn_l"], trials=[66],
                  missing={0: []}, passes=1),
}
for _i in range(12):                                   # enough subjects for the `--short` slice [11:12]
    SUBJECTS[f"tiny{_i:02d}"] = dict(seed=10 + _i, mass=60.0 + _i, bodies=["calcn_l", "calcn_r"], trials=[58 + _i],
                                     missing={0: []}, passes=2)

NUM_DOFS = 23
NUM_JOINTS = 12
ROOT_HISTORY_LEN = 10

# field -> (length given the subject, scale); lengths of the contact fields follow the SUBJECT's own body list
_FIELDS = [
    ("pos", lambda s: NUM_DOFS, 1.0), ("vel", lambda s: NUM_DOFS, 3.0), ("acc", lambda s: NUM_DOFS, 40.0),
    ("tau", lambda s: NUM_DOFS, 120.0),
    ("jointCentersInRootFrame", lambda s: 3 * NUM_JOINTS, 0.8),
    ("rootLinearVelInRootFrame", lambda s: 3, 1.5), ("rootAngularVelInRootFrame", lambda s: 3, 2.5),
    ("rootLinearAccInRootFrame", lambda s: 3, 9.0), ("rootAngularAccInRootFrame", lambda s: 3, 14.0),
    ("rootPosHistoryInRootFrame", lambda s: 3 * ROOT_HISTORY_LEN, 0.4),
    ("rootEulerHistoryInRootFrame", lambda s: 3 * ROOT_HISTORY_LEN, 0.2),
    ("residualWrenchInRootFrame", lambda s: 6, 30.0), ("comAccInRootFrame", lambda s: 3, 6.0),
    ("groundContactWrenchesInRootFrame", lambda s: 6 * len(s["bodies"]), 800.0),
    ("groundContactForceInRootFrame", lambda s: 3 * len(s["bodies"]), 900.0),
    ("groundContactCenterOfPressureInRootFrame", lambda s: 3 * len(s["bodies"]), 0.5),
    ("groundContactTorqueInRootFrame", lambda s: 3 * len(s["bodies"]), 70.0),
]


def field_value(spec: dict, trial: int, frame: int, pass_index: int, field_index: int, n: int, scale: float) -> np.ndarray:
    """float64 vector of one field of one processing pass of one frame (irrational-ish phases: the float32 rounding and
    the `/ mass` quotient are exercised on full mantissas)"""
    c = np.arange(n, dtype=np.float64)
    ph = 0.37 * c + 0.0113 * frame * (1.0 + 0.13 * field_index) + 1.7 * trial + 0.31 * spec["seed"] + 2.1 * pass_index \
        + 0.71 * field_index
    return scale * (np.sin(ph) + 0.25 * np.cos(2.3 * ph + 0.5))


class FramePass:
    def __init__(self, spec: dict, trial: int, frame: int, pass_index: int):
        for fi, (name, length, scale) in enumerate(_FIELDS):
            setattr(self, name, field_value(spec, trial, frame, pass_index, fi, length(spec), scale))


class Frame:
    def __init__(self, spec: dict, trial: int, frame: int):
        self.trial, self.t = trial, frame
        self.processingPasses: List[FramePass] = [FramePass(spec, trial, frame, p) for p in range(spec["passes"])]


class _BodyNode:
    def __init__(self, name: str):
        self._name = name

    def getName(self) -> str:
        return self._name


class _Dof:
    def __init__(self, name: str):
        self._name = name

    def getName(self) -> str:
        return self._name


class _Skeleton:
    """what `compute_report=True` calls on a skeleton (src/loss/RegressionLossEvaluator.py:265-286); the inverse dynamics
    is a closed form of its arguments -- enough to check the plumbing, not physics.  `dof_names`: what
    `inspect_dof_indices` walks (src/data/AddBiomechanicsDataset.py:141-156); the default is the same 23 names for every
    subject, a test may hand a skeleton another list"""

    def __init__(self, mass: float = 70.0, dof_names=None):
        self._mass, self._q, self._dq = mass, np.zeros(NUM_DOFS), np.zeros(NUM_DOFS)
        self._dofs = list(dof_names) if dof_names is not None else [f"dof_{j}" for j in range(NUM_DOFS)]

    def getBodyNode(self, name: str) -> _BodyNode:
        return _BodyNode(name)

    def getNumDofs(self) -> int:
        return len(self._dofs)

    def getDofByIndex(self, j: int) -> _Dof:
        return _Dof(self._dofs[j])

    def getMass(self) -> float:
        return self._mass

    def setPositions(self, q):
        self._q = np.asarray(q, dtype=np.float64)

    def setVelocities(self, dq):
        self._dq = np.asarray(dq, dtype=np.float64)

    def getInverseDynamicsFromPredictions(self, acc, bodies, wrenches, root_residual):
        w = sum(float(np.sum(x)) for x in wrenches)
        return self._mass * np.asarray(acc, dtype=np.float64) + 0.1 * self._q - 0.01 * self._dq + 1e-3 * w


class SubjectOnDisk:
    opened: List[str] = []          # every path a SubjectOnDisk was constructed on (worker re-open test)

    def __init__(self, path: str):
        name = os.path.splitext(os.path.basename(path))[0]
        if name not in SUBJECTS:
            raise RuntimeError(f"fake nimble: no subject {name!r}")
        self.path, self.spec = path, SUBJECTS[name]
        SubjectOnDisk.opened.append(path)

    def getNumDofs(self) -> int:
        return NUM_DOFS

    def getGroundForceBodies(self) -> List[str]:
        return list(self.spec["bodies"])

    def getNumTrials(self) -> int:
        return len(self.spec["trials"])

    def getTrialLength(self, trial: int) -> int:
        return self.spec["trials"][trial]

    def getMissingGRF(self, trial: int) -> List[MissingGRFReason]:
        out = [MissingGRFReason.notMissingGRF] * self.spec["trials"][trial]
        for a, b in self.spec["missing"].get(trial, []):
            for f in range(a, b + 1):
                out[f] = MissingGRFReason.unmeasuredExternalForceDetected if f % 2 else \
                    MissingGRFReason.measuredGrfZeroWhenAccelerationNonZero
        return out

    def getMassKg(self) -> float:
        return self.spec["mass"]

    def getNumProcessingPasses(self) -> int:
        return self.spec["passes"]

    def getTrialName(self, trial: int) -> str:
        return f"trial_{trial}"

    def readSkel(self, processing_pass: int, geometry_folder=None) -> _Skeleton:
        return _Skeleton(self.spec["mass"])

    def readFrames(self, trial: int, startFrame: int, numFramesToRead: int = 1, includeSensorData: bool = True,
                   includeProcessingPasses: bool = True, stride: int = 1, contactThreshold: float = 1.0) -> List[Frame]:
        n = self.spec["trials"][trial]
        out = []
        for k in range(numFramesToRead):
            f = startFrame + k * stride
            if f >= n:
                break
            out.append(Frame(self.spec, trial, f))
        return out


biomechanics = types.SimpleNamespace(SubjectOnDisk=SubjectOnDisk, MissingGRFReason=MissingGRFReason, Frame=Frame,
                                     FramePass=FramePass, FrameList=list)
dynamics = types.SimpleNamespace(Skeleton=_Skeleton, BodyNode=_BodyNode)


def make_tree(root: str, names=("alpha", "beta", "gamma")) -> List[str]:
    """Create empty .b3d files as ONE FILE PER NESTED DIRECTORY (root/alpha.b3d, root/d1/beta.b3d, root/d1/d2/gamma.b3d)
    so `os.walk` -- whose order inside a directory is the file system's -- visits them in `names` order everywhere.
    Also drops a file the loader must skip ('vander' in the name) and a non-.b3d file.  Returns the .b3d paths in order."""
    paths, d = [], root
    for i, name in enumerate(names):
        os.makedirs(d, exist_ok=True)
        p = os.path.join(d, name + ".b3d")
        open(p, "wb").close()
        paths.append(p)
        if i == 0:
            open(os.path.join(d, "notes.txt"), "wb").close()
        if i == 1:
            open(os.path.join(d, "VanDerZee2022_s01.b3d"), "wb").close()
        d = os.path.join(d, f"d{i + 1}")
    return paths
