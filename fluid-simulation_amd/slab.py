"""z-slab decomposition: the plane arithmetic of the library's slab path, restated for host code
(launchers, tests).  The reference is single-process (advection-reflection/main.cpp has no
decomposition); the cut is along z because z is the slowest axis of its index() (main.cpp:324), so
a slab is one contiguous range of planes.
"""
from dataclasses import dataclass

HALO = 2   # planes either side: one for the stencil, one more for backtraces of up to two cells
DEEP_HALO = 4   # planes either side that a pass of two sweeps reads (the library allocates them where every slab owns >= 4 planes)


def split_planes(nz, nparts, part):
    """Owned plane range [z0, z1) of `part`: nz // nparts planes each, the first nz % nparts one more."""
    base, rem = divmod(nz, nparts)
    z0 = part * base + min(part, rem)
    return z0, z0 + base + (1 if part < rem else 0)


@dataclass
class SlabLayout:
    nz: int
    nparts: int
    part: int
    halo: int = HALO

    def __post_init__(self):
        self.z0, self.z1 = split_planes(self.nz, self.nparts, self.part)
        self.halo = self.halo if self.nparts > 1 else 0
        if self.nparts > 1 and self.nz // self.nparts < self.halo:
            raise ValueError(f"too many slabs: every slab needs at least {self.halo} planes")
        self.gz0 = self.z0 - self.halo                 # global z of local plane 0
        self.nzl = (self.z1 - self.z0) + 2 * self.halo  # local planes, halo included
        self.zb = max(self.z0, 1) - self.gz0            # local range of the planes this slab relaxes
        self.ze = max(self.zb, min(self.z1, self.nz - 1) - self.gz0)
        self.has_lo = self.part > 0
        self.has_hi = self.part < self.nparts - 1
        self.lz_lo_interface = self.halo                          # my first owned plane
        self.lz_hi_interface = self.halo + (self.z1 - self.z0)    # my ghost copy of the upper slab's first plane

    def local_plane(self, gz):
        return gz - self.gz0


def take_parity(colour, i_am_upper):
    """After a half-sweep of `colour`, each side of an interface holds the fresh w of the interface
    plane on one checkerboard half: the upper slab where the cell over the interface has the colour,
    the lower slab on the rest.  Returns the parity (x+y+z) & 1 of the locations to take from the
    neighbour."""
    return (colour ^ 1) if i_am_upper else colour
