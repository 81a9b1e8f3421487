"""Host-side mirror of the reference's `resdomain` module (src/res_domain.f90) over the C-ABI.

Same names and argument meaning as the Fortran routines; integers only, no GPU needed.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import Region, ResSizes, check, ip

G4_OFF, G2_OFF, GP_OFF, GS_OFF, GT_OFF, G_SIZE = 0, 147456, 152064, 156672, 161280, 165888
XGRID, YGRID, ZGRID = 96, 48, 8


def processor_decomposition_manual(proc_number, numprocs, number_of_regions):
    """src/res_domain.f90:64-94 -> int32 array of the regions owned by `proc_number`."""
    buf = np.zeros(number_of_regions // numprocs + 1, dtype=np.int32)
    n = check(_lib.lib().sml_domain_decompose(proc_number, numprocs, number_of_regions, ip(buf), buf.size))
    return buf[:n].copy()


def initializedomain(num_regions, region_num, overlap=1, num_vert_levels=1, vert_level=1, vert_overlap=0):
    """src/res_domain.f90:96-121 -> Region (the grid_type extents)."""
    g = Region()
    check(_lib.lib().sml_domain_region(int(num_regions), int(region_num), overlap, num_vert_levels, vert_level, vert_overlap, C.byref(g)))
    return g


def allocate_res_sizes(grid, m=6000, deg=6, local_predictvars=4, logp_bool=True, precip_bool=True,
                       sst_bool_input=True, tisr_input_bool=True, ml_only=False):
    """Integer sizing of allocate_res_new (src/mod_reservoir.f90:80-180) + u(t) segment offsets (:1851-1885)."""
    s = ResSizes()
    check(_lib.lib().sml_domain_sizes(C.byref(grid), m, deg, local_predictvars, int(logp_bool), int(precip_bool),
                                      int(sst_bool_input), int(tisr_input_bool), int(ml_only), C.byref(s)))
    return s


def out_map(num_regions, region_num, num_vert_levels=1, vert_level=1, vert_overlap=0, precip_bool=True):
    """outvec element -> (index into G, mean/std slot); ordering of tile_full_grid_with_local_state_vec_res1d."""
    cap = 4 * XGRID * YGRID * ZGRID
    gi, si = np.zeros(cap, dtype=np.int32), np.zeros(cap, dtype=np.int32)
    n = check(_lib.lib().sml_domain_out_map(int(num_regions), int(region_num), num_vert_levels, vert_level, vert_overlap,
                                            int(precip_bool), ip(gi), ip(si), cap))
    return gi[:n].copy(), si[:n].copy()


def in_map(num_regions, region_num, overlap=1, num_vert_levels=1, vert_level=1, vert_overlap=0, precip_bool=True,
           sst_bool_input=True, tisr_input_bool=True):
    """input element -> (index into G, mean/std slot); ordering of tile_4d_and_logp_to_local_state_input + sst + tisr."""
    cap = 8 * XGRID * YGRID * ZGRID
    gi, si = np.zeros(cap, dtype=np.int32), np.zeros(cap, dtype=np.int32)
    n = check(_lib.lib().sml_domain_in_map(int(num_regions), int(region_num), overlap, num_vert_levels, vert_level, vert_overlap,
                                           int(precip_bool), int(sst_bool_input), int(tisr_input_bool), ip(gi), ip(si), cap))
    return gi[:n].copy(), si[:n].copy()


def target_map(num_regions, region_num, overlap=1, num_vert_levels=1, vert_level=1, vert_overlap=0, precip_bool=True):
    """tile_full_input_to_target_data (src/res_domain.f90:602-689) as 0-based positions into the region's input vector:
    targets = trainingdata[target_map, :]."""
    cap = 8 * XGRID * YGRID * ZGRID
    pos = np.zeros(cap, dtype=np.int32)
    n = check(_lib.lib().sml_domain_target_map(int(num_regions), int(region_num), overlap, num_vert_levels, vert_level, vert_overlap,
                                               int(precip_bool), ip(pos), cap))
    return pos[:n].copy()


def find_closest_divisor(target, number):
    """find_closest_divisor (src/mod_utilities.f90:1598-1636)"""
    return check(_lib.lib().sml_find_closest_divisor(int(target), int(number)))


def calendar_date(hours_elapsed, startyear=1981):
    """get_current_time_delta_hour (src/mod_calendar.f90:24-91): (year, month, day, hour)."""
    out = np.zeros(4, dtype=np.int32)
    check(_lib.lib().sml_calendar_date(int(startyear), int(hours_elapsed), ip(out)))
    return tuple(int(v) for v in out)


def tisr_index(hours_elapsed, startyear=1981):
    """get_tisr_by_date (src/mpires.f90:1676-1708): 1-based slice of the 8760-hour TISR table."""
    return check(_lib.lib().sml_tisr_index(int(startyear), int(hours_elapsed)))
