! Test support only: a module named `mpi` with the two calls the reference's program main makes itself (src/parallelmain.f90:275-277:
! MPI_Barrier, mpi_finalize), so that the reference's driver can be COMPILED IN PLACE and linked against the drop-in modules in an
! image without a Fortran MPI (tests/test_fortran_boundary.py).  A site build uses its MPI's own module; nothing here is shipped.
module mpi
  implicit none
contains
  subroutine MPI_Barrier(comm, ierr)
    integer :: comm, ierr
    ierr = 0
  end subroutine
  subroutine mpi_finalize(ierr)
    integer :: ierr
    ierr = 0
  end subroutine
end module mpi
