"""MI355X-native 3D MAC-grid fluid step (advection, reflection, red-black Gauss-Seidel projection).

The directory name contains a hyphen, so the package is imported through
`fsamd_loader.load()` at the repository root (module name `fluid_simulation_amd`).
"""
from .binding import *  # noqa: F401,F403
from .binding import FluidSim2D, FluidSim3D, FluidSimError, Config, Stats, load_library, default_config, comm_unique_id, EXPORTS, LIB_PATH
