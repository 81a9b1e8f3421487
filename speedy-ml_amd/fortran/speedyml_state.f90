! What the drop-in modules of one rank share: the bank that holds every reservoir of the rank in HBM (it replaces the per-reservoir MKL
! handles of reservoir_type), the hybrid engine behind mpires::sendrecievegrid, and the bookkeeping of the batched predict.
module speedyml_state
  use iso_c_binding
  implicit none
  type(c_ptr), save :: hip_bank = c_null_ptr, hip_engine = c_null_ptr
  integer, save :: hip_capacity = 0, hip_loaded = 0, hip_predicted = 0
  integer(c_int), allocatable, save :: region_of_slot(:), sst_input_of_slot(:)
  logical, allocatable, save :: slot_predicted(:)
  logical, save :: host_mirror = .false.
end module speedyml_state
