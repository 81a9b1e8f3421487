! What the drop-in modules of one rank share: the bank that holds every reservoir of the rank in HBM (it replaces the per-reservoir MKL
! handles of reservoir_type), the slab-ocean bank beside it (slot i of both = the rank's i-th region), the hybrid engine behind
! mpires::sendrecievegrid, the communicator of a multi-rank run and the bookkeeping of the batched predict calls.
module speedyml_state
  use iso_c_binding
  implicit none
  type(c_ptr), save :: hip_bank = c_null_ptr, hip_engine = c_null_ptr, hip_slab_bank = c_null_ptr, hip_comm = c_null_ptr
  integer, save :: hip_capacity = 0, hip_loaded = 0, hip_predicted = 0
  integer(c_int), allocatable, save :: region_of_slot(:), sst_input_of_slot(:)
  logical, allocatable, save :: slot_predicted(:)
  ! slab ocean: which slots hold a slab reservoir (sst_bool_prediction), how many, and the batched predict_slab_ml bookkeeping
  integer(c_int), allocatable, save :: slab_sea_of_slot(:)
  logical, allocatable, save :: slab_predicted(:)
  integer, save :: slab_loaded = 0, slab_done = 0, slab_max_d = 0, slab_max_out = 0
  logical, save :: slab_hybrid_ocean = .false.
  logical, save :: host_mirror = .false.
end module speedyml_state
