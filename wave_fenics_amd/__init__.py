"""wave_fenics_amd -- MI355X-native matrix-free operator engine for the explicit
RK wave-equation loop of Excalibur-SLE/wave-fenics.

The compute path is libwavehip.so (hand-written HIP for gfx950 behind the C ABI
of include/wavehip.h); this package is the host-side mirror of the reference's
operator interface (common/operators.hpp, common/cuda/*.hpp,
common/LinearGLL.hpp, demo/gpu_scatter_mpi/VectorUpdater.hpp) on top of it.
PyTorch is used for device memory, streams and torch.distributed only.
There is no CPU fallback: importing the operators without libwavehip raises.
"""
from ._lib import lib, check, WavehipError  # noqa: F401
from .box import BoxMesh, FunctionSpace, IndexMap, create_box, create_functionspace, lattice_numbering, renumber  # noqa: F401
from .operators import (  # noqa: F401
    StiffnessOperator, MassOperator, SpectralMassOperator, MassOperatorLumped,
    gather, scatter, transform1, tsmm, tabulate_gll, tabulate_dense, precompute_geometric_data,
    quadrature_1d, tabulate_1d, compute_geometry_rule,
)
from . import la  # noqa: F401
