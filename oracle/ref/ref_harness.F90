! ref_harness.F90 -- TEST INFRASTRUCTURE (oracle/_ref build only; never shipped, never on the product path).
!
! A caller written for this repo that drives the *unmodified* reference modules (compiled in place from
! /root/reference by oracle/ref/Makefile) through the tracer hot path and dumps raw golden vectors:
!   * init sequence of prim_init1/prim_init2 (reference: src/share/prim_driver_mod.F90:32-371,375-696)
!     minus namelist/IO/restart (those need netCDF/PIO, absent here),
!   * the prim_run_subcycle / prim_step loop (prim_driver_mod.F90:701-943),
!   * prescribed DCMIP fields filled the way dcmip_wrapper_mod.F90:49-266 does, by calling the reference's
!     own dcmip_123_mod point functions (dcmip_wrapper_mod itself cannot be built: it drags in
!     global_norms_mod -> mesh_mod -> netCDF).
! Everything numerical on the hot path (CubeTopology, cube_init_atomic, mass_matrix, derivinit,
! genEdgeSched, Prim_Advec_Tracers_remap, vertical_remap, remap_Q_ppm, divergence_sphere, ...) is the
! reference's code.
!
! stdin (list-directed): ne qsize nsteps tstep nu_q testcase(1|2) dumpfreq  /  'outdir'  /  'vcoord dir'
program ref_harness
  use kinds,              only : real_kind, longdouble_kind
  use dimensions_mod,     only : np, nlev, nlevp, ne, nelem, nelemd, nelemdmax, qsize, qsize_d, npart, &
                                 nnodes, nmpi_per_node
  use control_mod,        only : topology, partmethod, nu_q, limiter_option, hypervis_order, &
                                 hypervis_subcycle_q, rsplit, qsplit, test_case, cubed_sphere_map, &
                                 hypervis_power, hypervis_scaling, vert_remap_q_alg, integration, &
                                 tstep_type, nu, nu_p, nu_s
  use params_mod,         only : SFCURVE
  use parallel_mod,       only : parallel_t, initmp, iam, haltmp, global_shared_buf, nrepro_vars, &
                                 mpiinteger_t
  use hybrid_mod,         only : hybrid_t, hybrid_create
  use thread_mod,         only : nthreads
  use element_mod,        only : element_t, allocate_element_desc
  use gridgraph_mod,      only : gridvertex_t, gridedge_t, allocate_gridvertex_nbrs
  use metagraph_mod,      only : metavertex_t, localelemcount, initmetagraph
  use schedtype_mod,      only : schedule
  use schedule_mod,       only : genEdgeSched
  use spacecurve_mod,     only : genspacepart
  use cube_mod,           only : cubeedgecount, cubeelemcount, cubetopology, cube_init_atomic, &
                                 rotation_init_atomic, set_corner_coordinates, assign_node_numbers_to_elem
  use quadrature_mod,     only : quadrature_t, gausslobatto
  use mass_matrix_mod,    only : mass_matrix
  use repro_sum_mod,      only : repro_sum, repro_sum_defaultopts, repro_sum_setopts
  use physical_constants, only : dd_pi, p0, g, Rgas
  use hybvcoord_mod,      only : hvcoord_t, hvcoord_init
  use time_mod,           only : timelevel_t, timelevel_init, timelevel_update, timelevel_qdp, tstep
  use derivative_mod,     only : derivative_t, divergence_sphere, gradient_sphere, laplace_sphere_wk, &
                                 divergence_sphere_wk
  use filter_mod,         only : filter_t
  use prim_advection_mod, only : prim_advec_init1, prim_advec_init2, prim_advec_tracers_remap, &
                                 vertical_remap, deriv
  use vertremap_mod,      only : remap_q_ppm
  use dcmip_123_mod,      only : test1_advection_deformation, test1_advection_hadley
#if defined(TSE_HIP) && defined(_OPENMP)
  use omp_lib,            only : omp_get_thread_num
#endif
#ifdef TSE_HIP
  ! built by transport_se_amd/fortran/Makefile: prim_advection_mod is the reference's file compiled with
  ! -DUSE_CUDA_FORTRAN=1, and `cuda_mod` is transport_se_amd/fortran/cuda_mod_hip.F90 (the HIP library's Fortran seam)
  use cuda_mod,           only : cuda_mod_init, copy_qdp_h2d, copy_qdp_d2h, advec_tracers_remap_rk2_hip, hip_seam_report
#endif
  implicit none
#include <mpif.h>

  type (element_t), pointer :: elem(:)
  type (parallel_t)   :: par
  type (hybrid_t)     :: hybrid
  type (timelevel_t)  :: tl
  type (hvcoord_t)    :: hvcoord
  type (filter_t)     :: flt
  type (quadrature_t) :: gp
  type (GridVertex_t), target, allocatable :: GridVertex(:)
  type (GridEdge_t),   target, allocatable :: GridEdge(:)
  type (MetaVertex_t), target, allocatable :: MetaVertex(:)
  real(kind=real_kind), allocatable :: aratio(:,:)
  real(kind=real_kind) :: area(1), dt, dt_remap, nu_q_in, time
  real(kind=real_kind), parameter :: T0 = 300.d0
  real(kind=real_kind) :: Hs
  logical :: rs_ddpdd, rs_recompute
  real(kind=real_kind) :: rs_rel
  integer :: ne_in, qsize_in, nsteps, tcase, dumpfreq, nelem_edge
  integer :: ie, j, r, ierr, n0_qdp, np1_qdp, istep, isub, nsub
  integer :: nthr_h, ithr_h, nets_h, nete_h, n0q_h, np1q_h      ! TSE_HARNESS_THREADS: the time loop inside !$OMP PARALLEL, as prim_main.F90:143-162
  type (hybrid_t) :: hybrid_h
  integer(kind=8) :: c0, c1, crate
  character(len=256) :: outdir, vdir
  character(len=512) :: fname

  par = initmp()
  if (par%masterproc) then
     read(*,*) ne_in, qsize_in, nsteps, dt, nu_q_in, tcase, dumpfreq
     read(*,*) outdir
     read(*,*) vdir
  endif
  call MPI_Bcast(ne_in,   1, MPI_INTEGER, 0, par%comm, ierr)
  call MPI_Bcast(qsize_in,1, MPI_INTEGER, 0, par%comm, ierr)
  call MPI_Bcast(nsteps,  1, MPI_INTEGER, 0, par%comm, ierr)
  call MPI_Bcast(tcase,   1, MPI_INTEGER, 0, par%comm, ierr)
  call MPI_Bcast(dumpfreq,1, MPI_INTEGER, 0, par%comm, ierr)
  call MPI_Bcast(dt,      1, MPI_DOUBLE_PRECISION, 0, par%comm, ierr)
  call MPI_Bcast(nu_q_in, 1, MPI_DOUBLE_PRECISION, 0, par%comm, ierr)
  call MPI_Bcast(outdir, 256, MPI_CHARACTER, 0, par%comm, ierr)
  call MPI_Bcast(vdir,   256, MPI_CHARACTER, 0, par%comm, ierr)

  ! ---- what readnl would have set (test/dcmip1-1/dcmip1-1.nl + run scripts) ----
  ne = ne_in;  qsize = qsize_in;  tstep = dt
  topology = "cube";  partmethod = SFCURVE;  npart = par%nprocs
  nmpi_per_node = 1;  nnodes = npart;  nthreads = 1
  nthr_h = 1
#if defined(TSE_HIP) && defined(_OPENMP)
  call env_int('TSE_HARNESS_THREADS', nthr_h)    ! horizontal OpenMP threads of the rank (NThreads of ctl_nl)
  nthreads = max(1, nthr_h)
#endif
  nu = 0; nu_p = 0; nu_s = 0; nu_q = nu_q_in
  limiter_option = 8; hypervis_order = 2; hypervis_subcycle_q = 1
  hypervis_power = 0; hypervis_scaling = 0
  qsplit = 1; rsplit = 3; tstep_type = 1; integration = "explicit"
  cubed_sphere_map = 0
  ! optional namelist overrides through the environment (make_golden.py --alg2; the seam-guard tests of the drop-in harness):
  ! TSE_NL_VERT_REMAP_Q_ALG, TSE_NL_QSPLIT, TSE_NL_HYPERVIS_SUBCYCLE_Q, TSE_NL_HYPERVIS_POWER, TSE_NL_HYPERVIS_SCALING
  call env_int('TSE_NL_VERT_REMAP_Q_ALG', vert_remap_q_alg)
  call env_int('TSE_NL_QSPLIT', qsplit)
  call env_int('TSE_NL_HYPERVIS_SUBCYCLE_Q', hypervis_subcycle_q)
  call env_real('TSE_NL_HYPERVIS_POWER', hypervis_power)
  call env_real('TSE_NL_HYPERVIS_SCALING', hypervis_scaling)
  if (tcase == 1) then
     test_case = "dcmip1-1"
  else
     test_case = "dcmip1-2"
  endif
  Hs = Rgas*T0/g

  call repro_sum_defaultopts(repro_sum_use_ddpdd_out=rs_ddpdd, repro_sum_rel_diff_max_out=rs_rel, &
                             repro_sum_recompute_out=rs_recompute)
  call repro_sum_setopts(repro_sum_use_ddpdd_in=rs_ddpdd, repro_sum_rel_diff_max_in=rs_rel, &
                         repro_sum_recompute_in=rs_recompute, repro_sum_master=par%masterproc, &
                         repro_sum_logunit=6)

  ! ---- topology, partition, schedule (prim_init1) ----
  nelem      = CubeElemCount()
  nelem_edge = CubeEdgeCount()
  allocate(GridVertex(nelem)); allocate(GridEdge(nelem_edge))
  do j = 1, nelem
     call allocate_gridvertex_nbrs(GridVertex(j))
  enddo
  call CubeTopology(GridEdge, GridVertex)
  call genspacepart(GridEdge, GridVertex)
  allocate(MetaVertex(1)); allocate(Schedule(1))
  call initMetaGraph(iam, MetaVertex(1), GridVertex, GridEdge)
  nelemd = LocalElemCount(MetaVertex(1))
  call mpi_allreduce(nelemd, nelemdmax, 1, MPIinteger_t, MPI_MAX, par%comm, ierr)
  allocate(elem(nelemd))
  call allocate_element_desc(elem)
  call genEdgeSched(elem, iam, Schedule(1), MetaVertex(1))
  allocate(global_shared_buf(nelemd, nrepro_vars)); global_shared_buf = 0

  ! ---- geometry + mass matrix + area correction ----
  gp = gausslobatto(np)
  do ie = 1, nelemd
     call set_corner_coordinates(elem(ie))
  enddo
  call assign_node_numbers_to_elem(elem, GridVertex)
  do ie = 1, nelemd
     call cube_init_atomic(elem(ie), gp%points)
  enddo
  call mass_matrix(par, elem)
  allocate(aratio(nelemd,1))
  do ie = 1, nelemd
     aratio(ie,1) = sum(elem(ie)%mp(:,:)*elem(ie)%metdet(:,:))
  enddo
  call repro_sum(aratio, area, nelemd, nelemd, 1, commid=par%comm)
  area(1) = 4*dd_pi/area(1)
  deallocate(aratio)
  if (par%masterproc) write(*,'(a,f22.18)') ' area correction alpha = ', area(1)
  do ie = 1, nelemd
     call cube_init_atomic(elem(ie), gp%points, area(1))
     call rotation_init_atomic(elem(ie), "contravariant")
  enddo
  call mass_matrix(par, elem)
  do ie = 1, nelemd
     elem(ie)%derived%omega_p = 0
     elem(ie)%state%dp3d = 0
     elem(ie)%derived%eta_dot_dpdn = 0
     elem(ie)%derived%vn0 = 0
     elem(ie)%derived%divdp = 0
     elem(ie)%derived%divdp_proj = 0
     elem(ie)%state%Qdp = 0
     elem(ie)%state%Q = 0
     elem(ie)%state%v = 0
  enddo

  call Prim_Advec_Init1(par, nthreads)
  call TimeLevel_init(tl)
  hybrid = hybrid_create(par, 0, 1)

  ! ---- prim_init2 ----
  hvcoord = hvcoord_init(trim(vdir)//'/acme-72m.ascii', trim(vdir)//'/acme-72i.ascii', .false., &
                         par%masterproc, ierr)
  if (ierr /= 0) call haltmp('hvcoord_init failed')
  call Prim_Advec_Init2(hybrid)

  call set_fields(tl%n0, 0.0d0)
  call qdp_from_q()
#ifdef TSE_HIP
  call cuda_mod_init(elem, hybrid, deriv(0), hvcoord)     ! prim_driver_mod.F90:686-689
#endif

  if (dumpfreq >= 0) then
     call dump_static()
     call dump_ops()
     call TimeLevel_Qdp(tl, qsplit, n0_qdp, np1_qdp)
     call dump_state(0, n0_qdp, tl%n0)
  endif

  ! ---- time loop: nsteps tracer steps, remap every rsplit (prim_run_subcycle) ----
  nsub = nsteps / rsplit
  call system_clock(c0, crate)
  istep = 0
#if defined(TSE_HIP) && defined(_OPENMP)
  if (nthr_h > 1) then
     ! The reference runs its time loop inside !$OMP PARALLEL (prim_main.F90:143-162): every thread calls prim_run_subcycle with its own
     ! nets:nete and hybrid, and so reaches every cuda_mod entry; the seam's BARRIER / MASTER sections let the master act for 1:nelemd.
     ! Host-side work is split by element range; the shared time level is advanced by the master between barriers.  Only the final
     ! state is dumped (dumpfreq = 0 semantics).
     !$OMP PARALLEL NUM_THREADS(nthr_h) DEFAULT(SHARED) PRIVATE(ithr_h, nets_h, nete_h, hybrid_h, isub, r, n0q_h, np1q_h)
     ithr_h   = omp_get_thread_num()
     hybrid_h = hybrid_create(par, ithr_h, nthr_h)
     nets_h   = 1 + (nelemd*ithr_h)/nthr_h
     nete_h   = (nelemd*(ithr_h + 1))/nthr_h
     if (ithr_h > 0) call Prim_Advec_Init2(hybrid_h)     ! prim_init2 runs per thread in the reference (prim_main.F90:102-111): deriv(ithr)
     do isub = 1, nsub
        call TimeLevel_Qdp(tl, qsplit, n0q_h, np1q_h)
        call copy_qdp_h2d(elem, n0q_h)
        do r = 1, rsplit
           if (r > 1) then
              !$OMP BARRIER
              !$OMP MASTER
              call TimeLevel_update(tl, "leapfrog")
              !$OMP END MASTER
              !$OMP BARRIER
           endif
           call my_prim_step_range(nets_h, nete_h, hybrid_h, ithr_h)
           !$OMP MASTER
           istep = istep + 1
           !$OMP END MASTER
        enddo
        call TimeLevel_Qdp(tl, qsplit, n0q_h, np1q_h)
        call vertical_remap(hybrid_h, elem, hvcoord, dt*qsplit*rsplit, tl%np1, np1q_h, nets_h, nete_h)
        call copy_qdp_d2h(elem, np1q_h)
        !$OMP BARRIER
        !$OMP MASTER
        if (dumpfreq >= 0 .and. isub == nsub) call dump_state(istep, np1q_h, tl%np1)
        call TimeLevel_update(tl, "leapfrog")
        !$OMP END MASTER
        !$OMP BARRIER
     enddo
     !$OMP END PARALLEL
     nsub = 0     ! (the serial loop below has nothing left to do)
  endif
#endif
  do isub = 1, nsub
#ifdef TSE_HIP
     call TimeLevel_Qdp(tl, qsplit, n0_qdp, np1_qdp)       ! prim_driver_mod.F90:781-784
     call copy_qdp_h2d(elem, n0_qdp)
#endif
     do r = 1, rsplit
        if (r > 1) call TimeLevel_update(tl, "leapfrog")
        call my_prim_step()
        istep = istep + 1
        if (dumpfreq > 0) then
           if (mod(istep, dumpfreq) == 0 .and. r < rsplit) then
              call TimeLevel_Qdp(tl, qsplit, n0_qdp, np1_qdp)
#ifdef TSE_HIP
              call copy_qdp_d2h(elem, np1_qdp)
#endif
              call dump_state(istep, np1_qdp, tl%np1)
           endif
        endif
     enddo
     call TimeLevel_Qdp(tl, qsplit, n0_qdp, np1_qdp)
     dt_remap = dt*qsplit*rsplit
     if (dumpfreq > 0) then
#ifdef TSE_HIP
        call copy_qdp_d2h(elem, np1_qdp)
#endif
        if (istep == rsplit) call dump_state(-istep, np1_qdp, tl%np1)   ! pre-remap state of the first remap
     endif
     call vertical_remap(hybrid, elem, hvcoord, dt_remap, tl%np1, np1_qdp, 1, nelemd)
#ifdef TSE_HIP
     call copy_qdp_d2h(elem, np1_qdp)                      ! prim_driver_mod.F90:798-801
#endif
     if (dumpfreq > 0) then
        if (mod(istep, dumpfreq) == 0 .or. isub == nsub) call dump_state(istep, np1_qdp, tl%np1)
     else if (dumpfreq == 0 .and. isub == nsub) then
        call dump_state(istep, np1_qdp, tl%np1)
     endif
     call TimeLevel_update(tl, "leapfrog")
  enddo
  call system_clock(c1)
  if (par%masterproc) then
     write(*,'(a,i8,a,f12.4,a)') ' ref_harness: ', istep, ' tracer steps in ', dble(c1-c0)/dble(crate), ' s'
     write(*,'(a,es14.6)') ' ref_harness: tracer-DOF-steps/s = ', &
          dble(nelem)*np*np*nlev*qsize*istep / (dble(c1-c0)/dble(crate))
  endif
#ifdef TSE_HIP
  if (par%masterproc) call hip_seam_report(istep)
#endif
  call haltmp('ref_harness done')

contains

  subroutine env_int(name, v)
    character(len=*), intent(in) :: name
    integer, intent(inout) :: v
    character(len=64) :: txt
    integer :: stat
    call get_environment_variable(name, txt, status=stat)
    if (stat == 0 .and. len_trim(txt) > 0) read(txt, *) v
  end subroutine env_int
  subroutine env_real(name, v)
    character(len=*), intent(in) :: name
    real(kind=real_kind), intent(inout) :: v
    character(len=64) :: txt
    integer :: stat
    call get_environment_variable(name, txt, status=stat)
    if (stat == 0 .and. len_trim(txt) > 0) read(txt, *) v
  end subroutine env_real

  ! prim_step (prim_driver_mod.F90:856-943) + prim_advance_exp (prim_advance_mod.F90:62-152), ur_weights(1)=1
  subroutine my_prim_step()
    call my_prim_step_range(1, nelemd, hybrid, 0)
  end subroutine my_prim_step
  ! the same for one thread's element range e0:e1 (its hybrid, its derivative_t); the tracer call takes nets:nete as the reference's does
  subroutine my_prim_step_range(e0, e1, hyb, ithr)
    integer, intent(in) :: e0, e1, ithr
    type (hybrid_t), intent(in) :: hyb
    integer :: ie, k
    character(len=8) :: whole_step_env
    do ie = e0, e1
       elem(ie)%derived%eta_dot_dpdn = 0
       elem(ie)%derived%vn0 = 0
       elem(ie)%derived%omega_p = 0
       elem(ie)%derived%dp(:,:,:) = elem(ie)%state%dp3d(:,:,:,tl%n0)
    enddo
    call set_fields_range(tl%np1, tl%nstep*dt, e0, e1)
    do ie = e0, e1
       do k = 1, nlev
          elem(ie)%derived%vn0(:,:,1,k) = elem(ie)%derived%vn0(:,:,1,k) + &
               1.0d0*elem(ie)%state%v(:,:,1,k,tl%n0)*elem(ie)%derived%dp(:,:,k)
          elem(ie)%derived%vn0(:,:,2,k) = elem(ie)%derived%vn0(:,:,2,k) + &
               1.0d0*elem(ie)%state%v(:,:,2,k,tl%n0)*elem(ie)%derived%dp(:,:,k)
       enddo
    enddo
#ifdef TSE_HIP
    ! TSE_HARNESS_WHOLE_STEP=1: the one-call entry a maintainer would hook into Prim_Advec_Tracers_remap_rk2 (INTEGRATION.md)
    ! instead of the reference's per-stage hooks
    call get_environment_variable('TSE_HARNESS_WHOLE_STEP', whole_step_env)
    if (trim(whole_step_env) == '1') then
       call TimeLevel_Qdp(tl, qsplit, n0_qdp, np1_qdp)
       call advec_tracers_remap_rk2_hip(elem, dt*qsplit, n0_qdp, np1_qdp)
       return
    endif
#endif
    call Prim_Advec_Tracers_remap(elem, deriv(ithr), hvcoord, flt, hyb, dt*qsplit, tl, e0, e1)
  end subroutine my_prim_step_range

  ! what set_dcmip_1_1_fields / set_dcmip_1_2_fields + set_element_state + set_extra_tracers do
  ! (dcmip_wrapper_mod.F90:49-243).  omega_p is set to 0: the wrapper's cache_midpoint_values reads
  ! uninitialised locals for it (dcmip_wrapper_mod.F90:246-253) and the tracer result does not depend on it.
  subroutine set_fields(nt, time)
    integer, intent(in) :: nt
    real(kind=real_kind), intent(in) :: time
    call set_fields_range(nt, time, 1, nelemd)
  end subroutine set_fields
  subroutine set_fields_range(nt, time, e0, e1)
    integer, intent(in) :: nt, e0, e1
    real(kind=real_kind), intent(in) :: time
    real(kind=real_kind) :: T,phis,ps,u,v,w,p,z,rho,q(4),lon,lat,term
    real(kind=real_kind) :: p_i(np,np,nlevp), p_m(np,np,nlev), q_m(np,np,nlev,4)
    integer :: ie,i,j,k,qi
    do ie = e0, e1
       do k = 1, nlev
          z = Hs*log(1.0d0/hvcoord%etam(k))
          p = p0*hvcoord%etam(k)
          do j = 1, np; do i = 1, np
             lon = elem(ie)%spherep(i,j)%lon; lat = elem(ie)%spherep(i,j)%lat
             q = 0
             if (tcase == 1) then
                call test1_advection_deformation(time,lon,lat,p,z,1,u,v,w,T,phis,ps,rho,q(1),q(2),q(3),q(4))
             else
                call test1_advection_hadley(time,lon,lat,p,z,1,u,v,w,T,phis,ps,rho,q(1),q(2))
             endif
             elem(ie)%state%v(i,j,1,k,nt) = u
             elem(ie)%state%v(i,j,2,k,nt) = v
             p_m(i,j,k) = p
             q_m(i,j,k,:) = q
          enddo; enddo
       enddo
       do k = 1, nlevp
          z = Hs*log(1.0d0/hvcoord%etai(k))
          p = p0*hvcoord%etai(k)
          do j = 1, np; do i = 1, np
             lon = elem(ie)%spherep(i,j)%lon; lat = elem(ie)%spherep(i,j)%lat
             if (tcase == 1) then
                call test1_advection_deformation(time,lon,lat,p,z,1,u,v,w,T,phis,ps,rho,q(1),q(2),q(3),q(4))
             else
                call test1_advection_hadley(time,lon,lat,p,z,1,u,v,w,T,phis,ps,rho,q(1),q(2))
             endif
             p_i(i,j,k) = p
             elem(ie)%derived%eta_dot_dpdn(i,j,k) = -g*rho*w
          enddo; enddo
       enddo
       elem(ie)%state%dp3d(:,:,:,nt) = p_i(:,:,2:nlevp) - p_i(:,:,1:nlev)
       elem(ie)%state%ps_v(:,:,nt)   = p_i(:,:,nlevp)
       elem(ie)%derived%dp           = p_i(:,:,2:nlevp) - p_i(:,:,1:nlev)
       elem(ie)%derived%omega_p      = 0
       if (time == 0.0d0) then
          do qi = 1, min(4,qsize)
             elem(ie)%state%Q(:,:,:,qi) = q_m(:,:,:,qi)
          enddo
          do qi = 1, qsize
             if ((tcase == 1 .and. qi >= 5) .or. (tcase == 2 .and. qi /= 2)) then
                do j = 1, np; do i = 1, np
                   term = sin(9.*elem(ie)%spherep(i,j)%lon)*sin(9.*elem(ie)%spherep(i,j)%lat)
                   if (term < 0.) then
                      elem(ie)%state%Q(i,j,:,qi) = 0
                   else
                      elem(ie)%state%Q(i,j,:,qi) = 1
                   endif
                enddo; enddo
             endif
          enddo
       endif
    enddo
  end subroutine set_fields_range

  ! prim_init2: Qdp = Q*dp(hybrid coefficients, ps_v(n0))   (prim_driver_mod.F90:646-669)
  subroutine qdp_from_q()
    integer :: ie,i,j,k,q
    real(kind=real_kind) :: dp
    do ie = 1, nelemd
       do k = 1, nlev; do q = 1, qsize; do i = 1, np; do j = 1, np
          dp = ( hvcoord%hyai(k+1) - hvcoord%hyai(k) )*hvcoord%ps0 + &
               ( hvcoord%hybi(k+1) - hvcoord%hybi(k) )*elem(ie)%state%ps_v(i,j,tl%n0)
          elem(ie)%state%Qdp(i,j,k,q,1) = elem(ie)%state%Q(i,j,k,q)*dp
          elem(ie)%state%Qdp(i,j,k,q,2) = elem(ie)%state%Q(i,j,k,q)*dp
       enddo; enddo; enddo; enddo
    enddo
  end subroutine qdp_from_q

  subroutine open_out(stem, istep)
    character(len=*), intent(in) :: stem
    integer, intent(in) :: istep
    character(len=16) :: tag
    if (istep < 0) then
       write(tag,'(a,i6.6)') 'pre', -istep
    else
       write(tag,'(i6.6)') istep
    endif
    write(fname,'(a,a,a,a,a,a,i4.4,a)') trim(outdir), '/', trim(stem), '_', trim(tag), '_r', par%rank, '.bin'
    open(unit=31, file=trim(fname), form='unformatted', access='stream', status='replace')
  end subroutine open_out

  ! static data: Dvv, vertical coordinate, per-element metric terms and edge descriptors
  subroutine dump_static()
    integer :: ie, i, j
    integer :: rev(8)
    call open_out('static', 0)
    write(31) int(ne,4), int(nelem,4), int(nelemd,4), int(qsize,4), int(nlev,4), int(np,4), &
              int(par%rank,4), int(par%nprocs,4)
    write(31) area(1)
    write(31) deriv(0)%Dvv
    write(31) real(gp%points,8), real(gp%weights,8)
    write(31) hvcoord%hyai, hvcoord%hybi, hvcoord%hyam, hvcoord%hybm, hvcoord%ps0
    do ie = 1, nelemd
       rev = 0
       do j = 1, 8
          if (elem(ie)%desc%reverse(j)) rev(j) = 1
       enddo
       write(31) int(elem(ie)%GlobalId,4), int(elem(ie)%desc%putmapP(1:8),4), &
                 int(elem(ie)%desc%getmapP(1:8),4), int(rev,4), int(elem(ie)%vertex%face_number,4)
       write(31) ((elem(ie)%spherep(i,j)%lon, i=1,np), j=1,np)
       write(31) ((elem(ie)%spherep(i,j)%lat, i=1,np), j=1,np)
       write(31) elem(ie)%D, elem(ie)%Dinv, elem(ie)%metdet, elem(ie)%rmetdet, elem(ie)%mp, &
                 elem(ie)%spheremp, elem(ie)%rspheremp
    enddo
    write(31) int(Schedule(1)%ncycles,4), int(Schedule(1)%nSendCycles,4), int(Schedule(1)%nRecvCycles,4)
    do j = 1, Schedule(1)%nSendCycles
       write(31) int(Schedule(1)%SendCycle(j)%dest,4), int(Schedule(1)%SendCycle(j)%ptrP,4), &
                 int(Schedule(1)%SendCycle(j)%lengthP,4)
    enddo
    write(31) int(Schedule(1)%MoveCycle(1)%ptrP,4), int(Schedule(1)%MoveCycle(1)%lengthP,4)
    close(31)
  end subroutine dump_static

  ! single-call goldens of the public element-local operators and remap_Q_ppm on LCG inputs
  subroutine dump_ops()
    integer, parameter :: nq = 3
    real(kind=real_kind) :: s(np,np), v(np,np,2), r1(np,np), r2(np,np,2)
    real(kind=real_kind) :: Qdp(np,np,nlev,nq), Qin(np,np,nlev,nq), dp1(np,np,nlev), dp2(np,np,nlev), sh(np,np)
    integer :: ie, i, j, k, q, it
    integer(kind=8) :: seed
    if (par%rank /= 0) return
    seed = 12345
    call open_out('ops', 0)
    write(31) int(min(nelemd,6),4), int(nq,4)
    do ie = 1, min(nelemd,6)
       do j = 1, np; do i = 1, np
          s(i,j) = lcg(seed); v(i,j,1) = lcg(seed) - 0.5d0; v(i,j,2) = lcg(seed) - 0.5d0
       enddo; enddo
       write(31) int(ie,4), s, v
       r1 = divergence_sphere(v, deriv(0), elem(ie));        write(31) r1
       r2 = gradient_sphere(s, deriv(0), elem(ie)%Dinv);     write(31) r2
       r1 = divergence_sphere_wk(v, deriv(0), elem(ie));     write(31) r1
       r1 = laplace_sphere_wk(s, deriv(0), elem(ie), .true.); write(31) r1
    enddo
    ! remap_Q_ppm: source grid = smooth reference thickness perturbed by <= 20 %, same column mass on target
    do it = 1, 2
       do j = 1, np; do i = 1, np
          do k = 1, nlev
             dp2(i,j,k) = ( hvcoord%hyai(k+1) - hvcoord%hyai(k) )*hvcoord%ps0 + &
                          ( hvcoord%hybi(k+1) - hvcoord%hybi(k) )*hvcoord%ps0
             dp1(i,j,k) = dp2(i,j,k)*(1.0d0 + 0.4d0*(lcg(seed) - 0.5d0)*(it-1) + 0.02d0*(lcg(seed)-0.5d0))
          enddo
          sh(i,j) = sum(dp1(i,j,:))/sum(dp2(i,j,:))
          dp2(i,j,:) = dp2(i,j,:)*sh(i,j)
          do q = 1, nq; do k = 1, nlev
             if (q == 1) then
                Qdp(i,j,k,q) = dp1(i,j,k)*lcg(seed)
             else if (q == 2) then
                Qdp(i,j,k,q) = dp1(i,j,k)*(0.5d0 + 0.5d0*sin(0.3d0*k + i + 2*j))
             else
                Qdp(i,j,k,q) = dp1(i,j,k)
                if (k > 20 .and. k < 40) Qdp(i,j,k,q) = 0
             endif
          enddo; enddo
       enddo; enddo
       Qin = Qdp
       call remap_Q_ppm(Qdp, np, nq, dp1, dp2)
       write(31) dp1, dp2, Qin, Qdp
    enddo
    close(31)
  end subroutine dump_ops

  function lcg(seed) result(x)
    integer(kind=8), intent(inout) :: seed
    real(kind=real_kind) :: x
    seed = mod(seed*1103515245_8 + 12345_8, 2147483648_8)
    x = dble(seed)/2147483648.0d0
  end function lcg

  ! dynamic state after a tracer step (or after remap)
  subroutine dump_state(istep, nq, nt)
    integer, intent(in) :: istep, nq, nt
    integer :: ie
    call open_out('state', istep)
    write(31) int(istep,4), int(nq,4), int(nelemd,4), int(qsize,4)
    do ie = 1, nelemd
       write(31) elem(ie)%state%Qdp(:,:,:,1:qsize,nq)
       write(31) elem(ie)%derived%vn0, elem(ie)%derived%dp, elem(ie)%derived%divdp, &
                 elem(ie)%derived%divdp_proj, elem(ie)%derived%eta_dot_dpdn(:,:,1:nlev), &
                 elem(ie)%derived%omega_p, elem(ie)%state%dp3d(:,:,:,nt), elem(ie)%state%ps_v(:,:,nt)
    enddo
    close(31)
  end subroutine dump_state

end program ref_harness
