! cuda_mod_hip.F90 -- the Fortran half of the drop-in boundary.
!
! Provides a module named `cuda_mod` with the public names the reference's accelerator seam expects
! (reference src/share/cuda_mod.F90:57-64), implemented over the C ABI of include/transport_se_hip.h with
! ISO_C_BINDING.  Compiling the reference's prim_advection_mod.F90 with -DUSE_CUDA_FORTRAN=1 against THIS module
! makes its own hooks
!     euler_step      -> call euler_step_cuda(np1_qdp,n0_qdp,dt,elem,hvcoord,hybrid,deriv,nets,nete,DSSopt,rhs_multiplier)
!                                                                      (prim_advection_mod.F90:715-718)
!     qdp_time_avg    -> call qdp_time_avg_cuda(elem,rkstage,n0_qdp,np1_qdp,limiter_option,0d0,nets,nete)   (:653-656)
!     vertical_remap  -> call vertical_remap_cuda(elem,hvcoord,dt,np1,np1_qdp,nets,nete)                    (:1279-1282)
! land in the HIP library; the driver calls cuda_mod_init / copy_qdp_h2d / copy_qdp_d2h exactly where
! prim_driver_mod.F90:686-689,726-728,781-784,798-801 does.  Host code stays Fortran; elem(:) stays the host copy.
! One MPI rank drives one GPU.  Threading as in the CUDA seam (cuda_mod.F90:6-8,211-212,408-409,549-550,587-594): the entries are
! called by ALL horizontal OpenMP threads of the rank (the time loop runs inside !$OMP PARALLEL, prim_main.F90:143-162), each with
! its own nets:nete; every entry is  !$OMP BARRIER / !$OMP MASTER ... !$OMP END MASTER / !$OMP BARRIER  and the master thread acts for
! elements 1:nelemd (elem(:) is the rank's whole shared array).  Outside a parallel region, or built without -fopenmp, the
! directives are no-ops.
module cuda_mod
  use iso_c_binding
  use kinds,          only : real_kind
  use dimensions_mod, only : np, nlev, nlevp, nelemd, qsize, qsize_d
  use element_mod,    only : element_t
  use derivative_mod, only : derivative_t
  use hybvcoord_mod,  only : hvcoord_t
  use hybrid_mod,     only : hybrid_t
  use parallel_mod,   only : abortmp
  use control_mod,    only : nu_q, limiter_option, rsplit, qsplit, vert_remap_q_alg, hypervis_subcycle_q, hypervis_power, &
                             hypervis_scaling
  use schedtype_mod,  only : schedule
  implicit none
  private
#include <mpif.h>
  public :: cuda_mod_init, euler_step_cuda, qdp_time_avg_cuda, vertical_remap_cuda, copy_qdp_d2h, copy_qdp_h2d
  ! not in the reference's list: the whole tracer step in one device call (the fast path, INTEGRATION.md section 2)
  public :: advec_tracers_remap_rk2_hip
  ! wall time spent inside the seam (everything between entering a cuda_mod routine and returning to the host code), by entry
  public :: hip_seam_report
  real(kind=8), save :: t_seam(5) = 0d0   ! 1 copy_qdp_h2d, 2 copy_qdp_d2h, 3 euler_step_cuda/qdp_time_avg_cuda, 4 whole-step call, 5 vertical_remap_cuda
  integer(kind=8), save :: t_c0

  ! mirror of tse_init_args (include/transport_se_hip.h)
  type, bind(C) :: tse_init_args
     integer(c_int) :: nelemd, qsize, device
     real(c_double) :: nu_q
     integer(c_int) :: limiter_option, rsplit
     type(c_ptr)    :: Dvv, hyai, hybi
     real(c_double) :: ps0
     type(c_ptr)    :: Dinv;      integer(c_size_t) :: Dinv_stride
     type(c_ptr)    :: metdet;    integer(c_size_t) :: metdet_stride
     type(c_ptr)    :: rmetdet;   integer(c_size_t) :: rmetdet_stride
     type(c_ptr)    :: spheremp;  integer(c_size_t) :: spheremp_stride
     type(c_ptr)    :: rspheremp; integer(c_size_t) :: rspheremp_stride
     type(c_ptr)    :: putmapP, getmapP, reverse
     integer(c_int) :: nsend;  type(c_ptr) :: send_peer, send_ptrP, send_lengthP
     integer(c_int) :: nrecv;  type(c_ptr) :: recv_peer, recv_ptrP, recv_lengthP
     type(c_funptr) :: exchange; type(c_ptr) :: exchange_user
     integer(c_int) :: vert_remap_q_alg
  end type tse_init_args

  interface
     integer(c_int) function tse_init(ctx, args) bind(C, name='tse_init')
       import; type(c_ptr), intent(out) :: ctx; type(tse_init_args), intent(in) :: args
     end function
     integer(c_int) function tse_host_register(ctx, base, bytes) bind(C, name='tse_host_register')
       import; type(c_ptr), value :: ctx, base; integer(c_size_t), value :: bytes
     end function
     integer(c_int) function tse_comm_unique_id(id) bind(C, name='tse_comm_unique_id')
       import; type(c_ptr), value :: id
     end function
     integer(c_int) function tse_comm_abort(ctx) bind(C, name='tse_comm_abort')
       import; type(c_ptr), value :: ctx
     end function
     integer(c_int) function tse_comm_init(ctx, id, rank, nranks) bind(C, name='tse_comm_init')
       import; type(c_ptr), value :: ctx, id; integer(c_int), value :: rank, nranks
     end function
     integer(c_int) function tse_comm_precheck(ctx, rank, nranks) bind(C, name='tse_comm_precheck')
       import; type(c_ptr), value :: ctx; integer(c_int), value :: rank, nranks
     end function
     integer(c_int) function tse_copy_qdp_h2d(ctx, q, stride, qsize_d, nt) bind(C, name='tse_copy_qdp_h2d')
       import; type(c_ptr), value :: ctx, q; integer(c_size_t), value :: stride; integer(c_int), value :: qsize_d, nt
     end function
     integer(c_int) function tse_copy_qdp_d2h(ctx, q, stride, qsize_d, nt) bind(C, name='tse_copy_qdp_d2h')
       import; type(c_ptr), value :: ctx, q; integer(c_size_t), value :: stride; integer(c_int), value :: qsize_d, nt
     end function
     integer(c_int) function tse_set_derived(ctx, vn0, s0, dp, s1, eta, s2, omega, s3) bind(C, name='tse_set_derived')
       import; type(c_ptr), value :: ctx, vn0, dp, eta, omega; integer(c_size_t), value :: s0, s1, s2, s3
     end function
     integer(c_int) function tse_set_divdp(ctx, divdp, s0, divdp_proj, s1) bind(C, name='tse_set_divdp')
       import; type(c_ptr), value :: ctx, divdp, divdp_proj; integer(c_size_t), value :: s0, s1
     end function
     integer(c_int) function tse_get_derived(ctx, a, s1, b, s2, c, s3, d, s4, e, s5, f, s6) bind(C, name='tse_get_derived')
       import; type(c_ptr), value :: ctx, a, b, c, d, e, f; integer(c_size_t), value :: s1, s2, s3, s4, s5, s6
     end function
     integer(c_int) function tse_euler_step(ctx, np1_qdp, n0_qdp, dt, DSSopt, rhs) bind(C, name='tse_euler_step')
       import; type(c_ptr), value :: ctx; integer(c_int), value :: np1_qdp, n0_qdp, DSSopt, rhs; real(c_double), value :: dt
     end function
     integer(c_int) function tse_advec_tracers_remap_rk2(ctx, dt, n0_qdp, np1_qdp) bind(C, name='tse_advec_tracers_remap_rk2')
       import; type(c_ptr), value :: ctx; real(c_double), value :: dt; integer(c_int), value :: n0_qdp, np1_qdp
     end function
     integer(c_int) function tse_qdp_time_avg(ctx, rkstage, n0_qdp, np1_qdp) bind(C, name='tse_qdp_time_avg')
       import; type(c_ptr), value :: ctx; integer(c_int), value :: rkstage, n0_qdp, np1_qdp
     end function
     integer(c_int) function tse_vertical_remap(ctx, dt, np1_qdp) bind(C, name='tse_vertical_remap')
       import; type(c_ptr), value :: ctx; real(c_double), value :: dt; integer(c_int), value :: np1_qdp
     end function
     function tse_last_error() bind(C, name='tse_last_error') result(p)
       import; type(c_ptr) :: p
     end function
     integer(c_int) function tse_halo_layout(ctx, ns, nr) bind(C, name='tse_halo_layout')
       import; type(c_ptr), value :: ctx; integer(c_int), intent(out) :: ns, nr
     end function
     integer(c_int) function tse_halo_minmax_layout(ctx, sl, rl) bind(C, name='tse_halo_minmax_layout')
       import; type(c_ptr), value :: ctx, sl, rl
     end function
     ! HIP runtime (libamdhip64): staging copies of the packed slots for a non-GPU-aware MPI
     integer(c_int) function hipMemcpy(dst, src, nbytes, kind) bind(C, name='hipMemcpy')
       import; type(c_ptr), value :: dst, src; integer(c_size_t), value :: nbytes; integer(c_int), value :: kind
     end function
  end interface

  type(c_ptr), save :: ctx = c_null_ptr
  integer(c_int), allocatable, target, save :: putm(:,:), getm(:,:), revm(:,:)
  integer(c_int), allocatable, target, save :: speer(:), sptr(:), slen(:), rpeer(:), rptr(:), rlen(:)
  real(c_double), allocatable, target, save :: dvv_c(:,:), hyai_c(:), hybi_c(:)
  ! multi-rank halo exchange state (the bndry_exchangeV body, bndry_mod.F90:74-124, on staged host copies)
  integer, save :: x_comm = -1, x_nsend = 0, x_nrecv = 0
  integer(c_int), allocatable, target, save :: x_slen(:,:), x_rlen(:,:)     ! (slot, kind+1)
  real(c_double), allocatable, target, save :: x_hsend(:), x_hrecv(:)

contains

  subroutine check(rc, where)
    integer(c_int), intent(in) :: rc
    character(len=*), intent(in) :: where
    character(kind=c_char), pointer :: msg(:)
    character(len=400) :: text
    integer :: i
    if (rc == 0) return
    call c_f_pointer(tse_last_error(), msg, [400])
    text = ' '
    do i = 1, 400
       if (msg(i) == c_null_char) exit
       text(i:i) = msg(i)
    enddo
    call seam_abort(where//': '//trim(text))
  end subroutine check

  ! abortmp (parallel_mod.F90:274-287) writes its message to buffered stdout and then calls MPI_Abort, which loses it whenever
  ! stdout is a pipe or a file; the message goes to stderr first (and both units are flushed)
  ! and abortmp hands MPI_Abort an uninitialised error code (often 0: the launcher then reports success), so the job is ended
  ! here with a defined one; abortmp stays behind it as the reference's own route
  subroutine seam_abort(text)
    character(len=*), intent(in) :: text
    integer :: ierr
    write(0,'(a)') ' cuda_mod_hip: ABORTING WITH ERROR: '//text
    flush(0); flush(6)
    call MPI_Abort(MPI_COMM_WORLD, 1, ierr)
    call abortmp(text)
  end subroutine seam_abort

  subroutine tic()
    call system_clock(t_c0)
  end subroutine tic
  subroutine toc(i)
    integer, intent(in) :: i
    integer(kind=8) :: c1, rate
    call system_clock(c1, rate)
    t_seam(i) = t_seam(i) + dble(c1 - t_c0)/dble(rate)
  end subroutine toc
  subroutine hip_seam_report(nsteps)
    integer, intent(in) :: nsteps
    write(*,'(a,i6,a)') ' hip seam: wall seconds inside the cuda_mod entries over ', nsteps, ' tracer steps'
    write(*,'(a,5f10.4)') ' hip seam: copy_qdp_h2d copy_qdp_d2h euler_step+time_avg whole_step vertical_remap =', t_seam
    write(*,'(a,es14.6)') ' hip seam: tracer-DOF-steps/s of the seam alone = ', &
         dble(nelemd)*np*np*nlev*qsize*nsteps/max(sum(t_seam), 1d-30)
  end subroutine hip_seam_report

  integer(c_size_t) function stride_of(a, b)
    type(c_ptr), intent(in) :: a, b
    stride_of = transfer(b, 0_c_size_t) - transfer(a, 0_c_size_t)
  end function stride_of

  ! cuda_mod_init(elem,hybrid,deriv,hvcoord)   (called at the end of prim_init2, prim_driver_mod.F90:686-689)
  subroutine cuda_mod_init(elem, hybrid, deriv, hvcoord)
    type(element_t),    intent(in), target :: elem(:)
    type(hybrid_t),     intent(in) :: hybrid
    type(derivative_t), intent(in) :: deriv
    type(hvcoord_t),    intent(in) :: hvcoord
    type(tse_init_args) :: a
    integer :: ie, j, ns, nr, e2, ierr, rc_comm, rc_any
    integer(c_int) :: ncs, ncr
    character(len=16) :: xmode
    logical :: use_rccl
    character(kind=c_char), target, save :: comm_id(128)
    !$OMP BARRIER
    !$OMP MASTER
    ! What the device path does not implement is refused here, with the reference's own error route, instead of being silently
    ! replaced by the default behaviour: the tracer time levels assume qsplit = 1 (TimeLevel_Qdp, time_mod.F90:85-109), the
    ! hyperviscosity is the single constant-coefficient application of euler_step (prim_advection_mod.F90:796-826; no tracer
    ! subcycling, no variable / tensor coefficient: derivative_mod.F90:2438-2445), the remap is remap_Q_ppm with ghost-cell
    ! variant 0|1 or 2 (prim_advection_mod.F90:230-341).
    if (qsplit /= 1) call seam_abort('cuda_mod_init(hip): qsplit must be 1')
    if (hypervis_subcycle_q /= 1) call seam_abort('cuda_mod_init(hip): hypervis_subcycle_q must be 1')
    if (hypervis_power /= 0 .or. hypervis_scaling /= 0) call seam_abort('cuda_mod_init(hip): hypervis_power and hypervis_scaling must be 0')
    if (vert_remap_q_alg < 0 .or. vert_remap_q_alg > 2) call seam_abort('cuda_mod_init(hip): vert_remap_q_alg must be 0, 1 or 2')
    allocate(putm(8,nelemd), getm(8,nelemd), revm(8,nelemd))
    do ie = 1, nelemd
       putm(:,ie) = elem(ie)%desc%putmapP(1:8)
       getm(:,ie) = elem(ie)%desc%getmapP(1:8)
       do j = 1, 8
          revm(j,ie) = merge(1, 0, elem(ie)%desc%reverse(j))
       enddo
    enddo
    allocate(dvv_c(np,np), hyai_c(nlevp), hybi_c(nlevp))
    dvv_c = deriv%Dvv; hyai_c = hvcoord%hyai; hybi_c = hvcoord%hybi
    ! neighbour-rank slots exactly as genEdgeSched built them (schedule_mod.F90:36-239, schedtype_mod.F90:7-29)
    ns = schedule(1)%nSendCycles; nr = schedule(1)%nRecvCycles
    allocate(speer(max(ns,1)), sptr(max(ns,1)), slen(max(ns,1)), rpeer(max(nr,1)), rptr(max(nr,1)), rlen(max(nr,1)))
    do j = 1, ns
       speer(j) = schedule(1)%SendCycle(j)%dest - 1
       sptr(j)  = schedule(1)%SendCycle(j)%ptrP
       slen(j)  = schedule(1)%SendCycle(j)%lengthP
    enddo
    do j = 1, nr
       rpeer(j) = schedule(1)%RecvCycle(j)%source - 1
       rptr(j)  = schedule(1)%RecvCycle(j)%ptrP
       rlen(j)  = schedule(1)%RecvCycle(j)%lengthP
    enddo
    x_comm = hybrid%par%comm; x_nsend = ns; x_nrecv = nr
    e2 = min(2, nelemd)
    a%nelemd = nelemd; a%qsize = qsize; a%device = -1; a%nu_q = nu_q
    a%limiter_option = limiter_option; a%rsplit = rsplit
    a%Dvv = c_loc(dvv_c); a%hyai = c_loc(hyai_c); a%hybi = c_loc(hybi_c); a%ps0 = hvcoord%ps0
    a%Dinv = c_loc(elem(1)%Dinv);           a%Dinv_stride = stride_of(c_loc(elem(1)%Dinv), c_loc(elem(e2)%Dinv))
    a%metdet = c_loc(elem(1)%metdet);       a%metdet_stride = stride_of(c_loc(elem(1)%metdet), c_loc(elem(e2)%metdet))
    a%rmetdet = c_loc(elem(1)%rmetdet);     a%rmetdet_stride = stride_of(c_loc(elem(1)%rmetdet), c_loc(elem(e2)%rmetdet))
    a%spheremp = c_loc(elem(1)%spheremp);   a%spheremp_stride = stride_of(c_loc(elem(1)%spheremp), c_loc(elem(e2)%spheremp))
    a%rspheremp = c_loc(elem(1)%rspheremp); a%rspheremp_stride = stride_of(c_loc(elem(1)%rspheremp), c_loc(elem(e2)%rspheremp))
    a%putmapP = c_loc(putm); a%getmapP = c_loc(getm); a%reverse = c_loc(revm)
    a%nsend = ns; a%send_peer = c_loc(speer); a%send_ptrP = c_loc(sptr); a%send_lengthP = c_loc(slen)
    a%nrecv = nr; a%recv_peer = c_loc(rpeer); a%recv_ptrP = c_loc(rptr); a%recv_lengthP = c_loc(rlen)
    a%exchange = c_null_funptr; a%exchange_user = c_null_ptr
    a%vert_remap_q_alg = vert_remap_q_alg
    ! bndry_exchangeV: TSE_EXCHANGE=rccl (one rank per GPU) lets the library exchange the halo itself with RCCL send/recv on its
    ! own stream -- the communicator id travels over MPI once, below; otherwise the MPI body of tse_f_exchange is the callback
    call get_environment_variable('TSE_EXCHANGE', xmode)
    use_rccl = (trim(xmode) == 'rccl') .and. hybrid%par%nprocs > 1
    ! (with rccl the MPI body stays registered but dormant: the transport of last resort if the communicator cannot be built)
    if (ns + nr > 0) a%exchange = c_funloc(tse_f_exchange)
    call get_environment_variable('TSE_DEVICE_PER_RANK', xmode)
    if (trim(xmode) == '1') a%device = hybrid%par%rank        ! single node: MPI rank r drives GPU r
    call check(tse_init(ctx, a), 'cuda_mod_init')
    ! elem(:) is one contiguous allocation (prim_driver_mod.F90:221): page-lock it once, so that copy_qdp_h2d/d2h and the
    ! per-step derived fields are single 2-D DMAs straight out of / into the element structures
    if (nelemd > 1) then
       call check(tse_host_register(ctx, c_loc(elem(1)), stride_of(c_loc(elem(1)), c_loc(elem(2)))*int(nelemd,c_size_t)), &
                  'tse_host_register')
    endif
    if (use_rccl) then
       ! ncclCommInitRank is a blocking collective: every rank first checks what it can check alone, and only if ALL are ready
       ! does anyone enter it (a rank failing alone inside would strand the others in the bootstrap)
       rc_comm = tse_comm_precheck(ctx, int(hybrid%par%rank,c_int), int(hybrid%par%nprocs,c_int))
       call MPI_Allreduce(rc_comm, rc_any, 1, MPI_INTEGER, MPI_MAX, hybrid%par%comm, ierr)
       if (rc_any == 0) then
          if (hybrid%par%rank == 0) call check(tse_comm_unique_id(c_loc(comm_id)), 'tse_comm_unique_id')
          call MPI_Bcast(comm_id, 128, MPI_CHARACTER, 0, hybrid%par%comm, ierr)
          rc_comm = tse_comm_init(ctx, c_loc(comm_id), int(hybrid%par%rank,c_int), int(hybrid%par%nprocs,c_int))
          call MPI_Allreduce(rc_comm, rc_any, 1, MPI_INTEGER, MPI_MAX, hybrid%par%comm, ierr)
       endif
       if (rc_any /= 0) then     ! every rank leaves RCCL together: the halo goes through tse_f_exchange (MPI) instead
          call check(tse_comm_abort(ctx), 'tse_comm_abort')
          if (hybrid%par%rank == 0) write(*,'(a)') ' cuda_mod_hip: WARNING: RCCL communicator could not be initialised; halo exchange over MPI'
       endif
    endif
    if (ns + nr > 0) then
       allocate(x_slen(max(ns,1),2), x_rlen(max(nr,1),2))
       x_slen(1:ns,1) = slen(1:ns); x_rlen(1:nr,1) = rlen(1:nr)
       call check(tse_halo_minmax_layout(ctx, c_loc(x_slen(1,2)), c_loc(x_rlen(1,2))), 'tse_halo_minmax_layout')
       call check(tse_halo_layout(ctx, ncs, ncr), 'tse_halo_layout')
       ! largest message: max(qsize*nlev + nlev, 2*qpad*nlev) layers per column (the bounds arrays carry the tracer count rounded up
       ! to a multiple of 4, transport_se_hip.h)
       allocate(x_hsend(max(1,ncs)*max(qsize*nlev + nlev, 2*((qsize+3)/4*4)*nlev)), x_hrecv(max(1,ncr)*max(qsize*nlev + nlev, 2*((qsize+3)/4*4)*nlev)))
    endif
    !$OMP END MASTER
    !$OMP BARRIER
  end subroutine cuda_mod_init

  ! bndry_exchangeV (bndry_mod.F90:74-124) on the packed slots: MPICH here is not GPU-aware, so the slots are staged
  ! through host buffers; with a GPU-aware MPI the two hipMemcpy calls disappear and the device pointers go to MPI.
  integer(c_int) function tse_f_exchange(user, sendbuf, recvbuf, nlyr, kind) bind(C)
    type(c_ptr),    value :: user, sendbuf, recvbuf
    integer(c_int), value :: nlyr, kind
    integer :: i, off, ierr, nreq, ntot_s, ntot_r
    integer :: req(x_nsend + x_nrecv), stat(MPI_STATUS_SIZE, x_nsend + x_nrecv)
    tse_f_exchange = 1
    ntot_s = sum(x_slen(1:x_nsend, kind+1)); ntot_r = sum(x_rlen(1:x_nrecv, kind+1))
    if (ntot_s > 0) then
       if (hipMemcpy(c_loc(x_hsend), sendbuf, int(ntot_s,c_size_t)*nlyr*8_c_size_t, 2_c_int) /= 0) return
    endif
    nreq = 0; off = 0
    do i = 1, x_nrecv
       if (x_rlen(i,kind+1) > 0) then
          nreq = nreq + 1
          call MPI_Irecv(x_hrecv(off*nlyr + 1), x_rlen(i,kind+1)*nlyr, MPI_DOUBLE_PRECISION, rpeer(i), 10, x_comm, req(nreq), ierr)
       endif
       off = off + x_rlen(i,kind+1)
    enddo
    off = 0
    do i = 1, x_nsend
       if (x_slen(i,kind+1) > 0) then
          nreq = nreq + 1
          call MPI_Isend(x_hsend(off*nlyr + 1), x_slen(i,kind+1)*nlyr, MPI_DOUBLE_PRECISION, speer(i), 10, x_comm, req(nreq), ierr)
       endif
       off = off + x_slen(i,kind+1)
    enddo
    call MPI_Waitall(nreq, req, stat, ierr)
    if (ntot_r > 0) then
       if (hipMemcpy(recvbuf, c_loc(x_hrecv), int(ntot_r,c_size_t)*nlyr*8_c_size_t, 1_c_int) /= 0) return
    endif
    tse_f_exchange = 0
  end function tse_f_exchange

  integer(c_size_t) function estride(elem)
    type(element_t), intent(in), target :: elem(:)
    estride = stride_of(c_loc(elem(1)%state%Qdp), c_loc(elem(min(2,size(elem)))%state%Qdp))
  end function estride

  subroutine copy_qdp_h2d(elem, nt)
    type(element_t), intent(in), target :: elem(:)
    integer, intent(in) :: nt
    !$OMP BARRIER
    !$OMP MASTER
    call tic()
    call check(tse_copy_qdp_h2d(ctx, c_loc(elem(1)%state%Qdp), estride(elem), int(qsize_d,c_int), int(nt,c_int)), 'copy_qdp_h2d')
    call toc(1)
    !$OMP END MASTER
    !$OMP BARRIER
  end subroutine copy_qdp_h2d

  subroutine copy_qdp_d2h(elem, nt)
    type(element_t), intent(in), target :: elem(:)
    integer, intent(in) :: nt
    !$OMP BARRIER
    !$OMP MASTER
    call tic()
    call check(tse_copy_qdp_d2h(ctx, c_loc(elem(1)%state%Qdp), estride(elem), int(qsize_d,c_int), int(nt,c_int)), 'copy_qdp_d2h')
    call toc(2)
    !$OMP END MASTER
    !$OMP BARRIER
  end subroutine copy_qdp_d2h

  subroutine euler_step_cuda(np1_qdp, n0_qdp, dt, elem, hvcoord, hybrid, deriv, nets, nete, DSSopt, rhs_multiplier)
    integer,              intent(in)            :: np1_qdp, n0_qdp
    real(kind=real_kind), intent(in)            :: dt
    type(element_t),      intent(inout), target :: elem(:)
    type(hvcoord_t),      intent(in)            :: hvcoord
    type(hybrid_t),       intent(in)            :: hybrid
    type(derivative_t),   intent(in)            :: deriv
    integer,              intent(in)            :: nets, nete, DSSopt, rhs_multiplier
    integer(c_size_t) :: s
    type(c_ptr) :: pdiv, peta, pomg
    ! (every thread arrives with its own nets:nete: the barrier makes the host-side writes of all of them -- elem%derived of the
    ! tracer step -- visible before the master stages them, and the closing barrier keeps the others out of Qdp until it is back)
    !$OMP BARRIER
    !$OMP MASTER
    call tic()
    s = estride(elem)   ! every field lives in the same fixed-size element_t, so one stride serves all
    ! The CUDA seam stages elem%derived on every call (cuda_mod.F90:535-547, 564-586).  Nothing on the host changes vn0, dp,
    ! divdp, eta_dot_dpdn or omega_p between the three euler_step calls of a tracer step (prim_advection_mod.F90:614-637), and
    ! the DSS'd divdp_proj / eta_dot_dpdn of the earlier stages are already on the device, so the inputs are uploaded once per
    ! tracer step (with the first stage, rhs_multiplier = 0) and only the variable this stage DSSes is copied back.
    if (rhs_multiplier == 0) then
       call check(tse_set_derived(ctx, c_loc(elem(1)%derived%vn0), s, c_loc(elem(1)%derived%dp), s, &
                                  c_loc(elem(1)%derived%eta_dot_dpdn), s, c_loc(elem(1)%derived%omega_p), s), 'tse_set_derived')
       call check(tse_set_divdp(ctx, c_loc(elem(1)%derived%divdp), s, c_loc(elem(1)%derived%divdp_proj), s), 'tse_set_divdp')
    endif
    call check(tse_euler_step(ctx, int(np1_qdp,c_int), int(n0_qdp,c_int), dt, int(DSSopt,c_int), int(rhs_multiplier,c_int)), &
               'euler_step_cuda')
    pdiv = c_null_ptr; peta = c_null_ptr; pomg = c_null_ptr
    if (DSSopt == 3) pdiv = c_loc(elem(1)%derived%divdp_proj)   ! DSSdiv_vdp_ave (prim_advection_mod.F90:454-456)
    if (DSSopt == 1) peta = c_loc(elem(1)%derived%eta_dot_dpdn)
    if (DSSopt == 2) pomg = c_loc(elem(1)%derived%omega_p)
    call check(tse_get_derived(ctx, pdiv, s, peta, s, pomg, s, c_null_ptr, 0_c_size_t, c_null_ptr, 0_c_size_t, &
                               c_null_ptr, 0_c_size_t), 'tse_get_derived')
    call toc(3)
    !$OMP END MASTER
    !$OMP BARRIER
  end subroutine euler_step_cuda

  ! The body of Prim_Advec_Tracers_remap_rk2 (prim_advection_mod.F90:600-636: divdp = div(vn0), three euler_steps, qdp_time_avg)
  ! as ONE call.  Only this entry may leave Qdp(np1) un-materialised between the stages (DSS on read), which makes it ~20 %
  ! faster than the three per-stage hooks.  The reference has no hook at this level; a maintainer adds, after
  ! `call TimeLevel_Qdp(tl, qsplit, n0_qdp, np1_qdp)` (prim_advection_mod.F90:604):
  !     #if USE_CUDA_FORTRAN
  !       call advec_tracers_remap_rk2_hip(elem, dt, n0_qdp, np1_qdp); call t_stopf('prim_advec_tracers_remap_rk2'); return
  !     #endif
  subroutine advec_tracers_remap_rk2_hip(elem, dt, n0_qdp, np1_qdp)
    type(element_t),      intent(inout), target :: elem(:)
    real(kind=real_kind), intent(in)            :: dt
    integer,              intent(in)            :: n0_qdp, np1_qdp
    integer(c_size_t) :: s
    !$OMP BARRIER
    !$OMP MASTER
    call tic()
    s = estride(elem)
    call check(tse_set_derived(ctx, c_loc(elem(1)%derived%vn0), s, c_loc(elem(1)%derived%dp), s, &
                               c_loc(elem(1)%derived%eta_dot_dpdn), s, c_loc(elem(1)%derived%omega_p), s), 'tse_set_derived')
    call check(tse_advec_tracers_remap_rk2(ctx, dt, int(n0_qdp,c_int), int(np1_qdp,c_int)), 'advec_tracers_remap_rk2_hip')
    ! what the three stages leave in elem%derived: DSS'd divdp_proj, eta_dot_dpdn, omega_p, and divdp
    call check(tse_get_derived(ctx, c_loc(elem(1)%derived%divdp_proj), s, c_loc(elem(1)%derived%eta_dot_dpdn), s, &
                               c_loc(elem(1)%derived%omega_p), s, c_loc(elem(1)%derived%divdp), s, c_null_ptr, 0_c_size_t, &
                               c_null_ptr, 0_c_size_t), 'tse_get_derived')
    call toc(4)
    !$OMP END MASTER
    !$OMP BARRIER
  end subroutine advec_tracers_remap_rk2_hip

  subroutine qdp_time_avg_cuda(elem, rkstage, n0_qdp, np1_qdp, limiter_option, nu_p, nets, nete)
    type(element_t),      intent(inout) :: elem(:)
    real(kind=real_kind), intent(in)    :: nu_p
    integer,              intent(in)    :: rkstage, n0_qdp, np1_qdp, nets, nete, limiter_option
    !$OMP BARRIER
    !$OMP MASTER
    call tic()
    call check(tse_qdp_time_avg(ctx, int(rkstage,c_int), int(n0_qdp,c_int), int(np1_qdp,c_int)), 'qdp_time_avg_cuda')
    call toc(3)
    !$OMP END MASTER
    !$OMP BARRIER
  end subroutine qdp_time_avg_cuda

  ! call site: vertical_remap_cuda(elem,hvcoord,dt,np1,np1_qdp,nets,nete)   (prim_advection_mod.F90:1280)
  subroutine vertical_remap_cuda(elem, hvcoord, dt, np1, np1_qdp, nets, nete)
    type(element_t),      intent(inout), target :: elem(:)
    type(hvcoord_t),      intent(in)    :: hvcoord
    real(kind=real_kind), intent(in)    :: dt
    integer,              intent(in)    :: np1, np1_qdp, nets, nete
    integer(c_size_t) :: s
    integer :: ie
    real(kind=real_kind), allocatable, target :: dp3d(:,:,:,:), psv(:,:,:)
    !$OMP BARRIER
    !$OMP MASTER
    call tic()
    s = estride(elem)
    call check(tse_set_derived(ctx, c_null_ptr, 0_c_size_t, c_loc(elem(1)%derived%dp), s, c_null_ptr, 0_c_size_t, &
                               c_null_ptr, 0_c_size_t), 'tse_set_derived')
    call check(tse_set_divdp(ctx, c_null_ptr, 0_c_size_t, c_loc(elem(1)%derived%divdp_proj), s), 'tse_set_divdp')
    call check(tse_vertical_remap(ctx, dt, int(np1_qdp,c_int)), 'vertical_remap_cuda')
    ! state%dp3d(:,:,:,np1) and state%ps_v(:,:,np1) are time-level slices: stage through dense arrays
    allocate(dp3d(np,np,nlev,nelemd), psv(np,np,nelemd))
    call check(tse_get_derived(ctx, c_null_ptr, 0_c_size_t, c_null_ptr, 0_c_size_t, c_null_ptr, 0_c_size_t, c_null_ptr, &
                               0_c_size_t, c_loc(dp3d), int(np*np*nlev*8,c_size_t), c_loc(psv), int(np*np*8,c_size_t)), &
               'tse_get_derived')
    do ie = 1, nelemd
       elem(ie)%state%dp3d(:,:,:,np1) = dp3d(:,:,:,ie)
       elem(ie)%state%ps_v(:,:,np1)   = psv(:,:,ie)
    enddo
    call toc(5)
    !$OMP END MASTER
    !$OMP BARRIER
  end subroutine vertical_remap_cuda

end module cuda_mod
