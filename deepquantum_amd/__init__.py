"""deepquantum_amd -- MI355X-native statevector backend with the QubitCircuit API of
TuringQ/deepquantum (``import deepquantum_amd as dq``).

Only the statevector hot path is implemented: gate application, expectation /
measurement reductions, ``get_unitary`` and the index-bit-sharded multi-GPU variant, all on hand-written
gfx950 HIP kernels behind the C ABI of ``include/dq_hip.h``.  Tensors must live on the GPU.
"""

__version__ = '0.1.0'

import torch

from . import backend, bitmath, executor, fusion, ops, qmath
from .adjoint import AdjointExpectation
from .circuit import DistributedQubitCircuit, QubitCircuit
from .communication import (
    cleanup_distributed, comm_exchange_arrays, comm_get_rank, comm_get_world_size, setup_distributed,
)
from .ansatz import (
    HHL, Ansatz, ControlledMultiplier, ControlledUa, NumberEncoder, PhiAdder, PhiModularAdder,
    QuantumConvolutionalNeuralNetwork, QuantumFourierTransform, QuantumPhaseEstimationSingleQubit, RandomCircuitG3, ShorCircuit, ShorCircuitFor15,
)
from .channel import (
    AmplitudeDamping, BitFlip, Depolarizing, GeneralizedAmplitudeDamping, Pauli, PhaseDamping, PhaseFlip,
)
from .gate import (
    CNOT, ArbitraryGate, Barrier, CombinedSingleGate, DoubleControlGate, DoubleGate, Fredkin, Hadamard,
    HamiltonianGate, Identity, ImaginarySwap, LatentGate, ParametricDoubleGate, ParametricSingleGate, PauliX, PauliY,
    PauliZ, PhaseShift, ProjectionJ, ReconfigurableBeamSplitter, Reset, Rx, Rxx, Rxy, Ry, Ryy, Rz, Rzz, SDaggerGate, SGate,
    SingleGate, Swap, TDaggerGate, TGate, Toffoli, TripleGate, U3Gate, UAnyGate,
)
from .layer import (
    CnotLayer, CnotRing, DoubleLayer, HLayer, Observable, ParametricSingleLayer, RxLayer, RyLayer, RzLayer, SingleLayer,
    U3Layer, XLayer, YLayer, ZLayer,
)
from .operation import Channel, Gate, Layer, Operation
from .qmath import amplitude_encoding, expectation, measure, multi_kron
from .state import DistributedQubitState, QubitState
from .utils import CapturedGraph, dtype_map
from . import qasm3  # noqa: E402  (after the circuit classes it builds on)
from .qasm3 import cir_to_qasm3, qasm3_to_cir
