!-----------------------------------------------------------------------
! pigs_capi -- ISO_C_BINDING interface to libpigs_hip.so (include/pigs_hip.h).
!
! This is the Fortran side of the drop-in boundary: a host written in
! Fortran 90 (like the reference) passes its own arrays -- Path(dim,Np,0:2*Nb),
! VTable(0:Nmax+1), LogWF(0:Nmax+1), all real(8), column-major -- unchanged.
! Particle indices are 1-based and bead indices 0-based exactly as in the
! reference; walker indices are 0-based (walkers are new).
! Every function returns 0 on success or a negative pigs_status.
!-----------------------------------------------------------------------
module pigs_capi

  use iso_c_binding
  implicit none

  integer(c_int), parameter :: PIGS_OK = 0

  ! mirrors `struct pigs_params` (the reference's module globals, global_mod.f90:5-12,
  ! system_mod.f90:8-9, plus dt)
  ! mirrors `struct pigs_sweep_params` (device-resident sampler, K6)
  type, bind(C) :: pigs_sweep_params
     integer(c_int32_t) :: Nlev, Nstag, CMFreq, Lstag
     real(c_double)     :: delta_cm
     real(c_double)     :: CWorm, density, rbin
     integer(c_int32_t) :: swapping, Nobdm, Nbin, Npw
     integer(c_int32_t) :: sampling = 0, reserved = 0      ! 0 = 'bis', 1 = 'sta'
  end type pigs_sweep_params

  type, bind(C) :: pigs_params
     integer(c_int32_t) :: dim, Np, Nb, Nmax
     integer(c_int32_t) :: trap, wf_table, v_table, reserved
     real(c_double)     :: dr, rcut2, dt, Rm
     real(c_double)     :: Lbox(3)
     real(c_double)     :: a_ho(3)
  end type pigs_params

  interface

     function pigs_ctx_create(p,VTable,LogWF,n_walkers,device_id,ctx) bind(C,name='pigs_ctx_create') result(rc)
       import :: c_int, c_int32_t, c_double, c_ptr, pigs_params
       type(pigs_params), intent(in) :: p
       real(c_double), intent(in)    :: VTable(*),LogWF(*)
       integer(c_int32_t), value     :: n_walkers,device_id
       type(c_ptr), intent(out)      :: ctx
       integer(c_int) :: rc
     end function pigs_ctx_create

     function pigs_ctx_destroy(ctx) bind(C,name='pigs_ctx_destroy') result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int) :: rc
     end function pigs_ctx_destroy

     function pigs_last_error() bind(C,name='pigs_last_error') result(msg)
       import :: c_ptr
       type(c_ptr) :: msg
     end function pigs_last_error

     function pigs_device_count(n) bind(C,name='pigs_device_count') result(rc)
       import :: c_int, c_int32_t
       integer(c_int32_t), intent(out) :: n
       integer(c_int) :: rc
     end function pigs_device_count

     function pigs_sync(ctx) bind(C,name='pigs_sync') result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int) :: rc
     end function pigs_sync

     function pigs_set_tuning(ctx,key,value) bind(C,name='pigs_set_tuning') result(rc)
       import :: c_int, c_int32_t, c_ptr, c_char
       type(c_ptr), value :: ctx
       character(kind=c_char), intent(in) :: key(*)
       integer(c_int32_t), value :: value
       integer(c_int) :: rc
     end function pigs_set_tuning

     function pigs_build_tables(Nmax,Rm,rmax,VTable,LogWF,dr_out) bind(C,name='pigs_build_tables') result(rc)
       import :: c_int, c_int32_t, c_double
       integer(c_int32_t), value :: Nmax
       real(c_double), value     :: Rm,rmax
       real(c_double)            :: VTable(*),LogWF(*),dr_out
       integer(c_int) :: rc
     end function pigs_build_tables

     function pigs_build_tables_kind(kind,Nmax,Rm,rmax,VTable,LogWF,dr_out) &
          & bind(C,name='pigs_build_tables_kind') result(rc)
       import :: c_int, c_int32_t, c_double
       integer(c_int32_t), value :: kind,Nmax
       real(c_double), value     :: Rm,rmax
       real(c_double)            :: VTable(*),LogWF(*),dr_out
       integer(c_int) :: rc
     end function pigs_build_tables_kind

     function pigs_path_upload(ctx,walker,Path) bind(C,name='pigs_path_upload') result(rc)
       import :: c_int, c_int32_t, c_double, c_ptr
       type(c_ptr), value        :: ctx
       integer(c_int32_t), value :: walker
       real(c_double), intent(in) :: Path(*)
       integer(c_int) :: rc
     end function pigs_path_upload

     function pigs_path_download(ctx,walker,Path) bind(C,name='pigs_path_download') result(rc)
       import :: c_int, c_int32_t, c_double, c_ptr
       type(c_ptr), value        :: ctx
       integer(c_int32_t), value :: walker
       real(c_double)            :: Path(*)
       integer(c_int) :: rc
     end function pigs_path_download

     function pigs_path_upload_all(ctx,Paths) bind(C,name='pigs_path_upload_all') result(rc)
       import :: c_int, c_double, c_ptr
       type(c_ptr), value         :: ctx
       real(c_double), intent(in) :: Paths(*)
       integer(c_int) :: rc
     end function pigs_path_upload_all

     function pigs_path_download_all(ctx,Paths) bind(C,name='pigs_path_download_all') result(rc)
       import :: c_int, c_double, c_ptr
       type(c_ptr), value :: ctx
       real(c_double)     :: Paths(*)
       integer(c_int) :: rc
     end function pigs_path_download_all

     ! replaces `call UpdateAction(...)` (reference vpi_mod.f90:2491), batched
     function pigs_delta_action_batch(ctx,n_items,walker,ip,ib,xnew,xold,DeltaS) &
          & bind(C,name='pigs_delta_action_batch') result(rc)
       import :: c_int, c_int32_t, c_int64_t, c_double, c_ptr
       type(c_ptr), value             :: ctx
       integer(c_int64_t), value      :: n_items
       integer(c_int32_t), intent(in) :: walker(*),ip(*),ib(*)
       real(c_double), intent(in)     :: xnew(*),xold(*)
       real(c_double)                 :: DeltaS(*)
       integer(c_int) :: rc
     end function pigs_delta_action_batch

     ! pinned, device-mapped staging arrays owned by the library (low-latency sampler path)
     function pigs_stage_reserve(ctx,capacity,keep,walker,ip,ib,xnew,xold,DeltaS) &
          & bind(C,name='pigs_stage_reserve') result(rc)
       import :: c_int, c_int64_t, c_ptr
       type(c_ptr), value        :: ctx
       integer(c_int64_t), value :: capacity,keep
       type(c_ptr), intent(out)  :: walker,ip,ib,xnew,xold,DeltaS
       integer(c_int) :: rc
     end function pigs_stage_reserve

     function pigs_delta_action_staged(ctx,n_items) bind(C,name='pigs_delta_action_staged') result(rc)
       import :: c_int, c_int64_t, c_ptr
       type(c_ptr), value        :: ctx
       integer(c_int64_t), value :: n_items
       integer(c_int) :: rc
     end function pigs_delta_action_staged

     function pigs_commit_reserve(ctx,capacity,keep,walker,ip,ib,x) bind(C,name='pigs_commit_reserve') result(rc)
       import :: c_int, c_int64_t, c_ptr
       type(c_ptr), value        :: ctx
       integer(c_int64_t), value :: capacity,keep
       type(c_ptr), intent(out)  :: walker,ip,ib,x
       integer(c_int) :: rc
     end function pigs_commit_reserve

     function pigs_commit_staged(ctx,n) bind(C,name='pigs_commit_staged') result(rc)
       import :: c_int, c_int64_t, c_ptr
       type(c_ptr), value        :: ctx
       integer(c_int64_t), value :: n
       integer(c_int) :: rc
     end function pigs_commit_staged

     function pigs_delta_action_parts(ctx,n_items,walker,ip,ib,xnew,xold,parts) &
          & bind(C,name='pigs_delta_action_parts') result(rc)
       import :: c_int, c_int32_t, c_int64_t, c_double, c_ptr
       type(c_ptr), value             :: ctx
       integer(c_int64_t), value      :: n_items
       integer(c_int32_t), intent(in) :: walker(*),ip(*),ib(*)
       real(c_double), intent(in)     :: xnew(*),xold(*)
       real(c_double)                 :: parts(*)
       integer(c_int) :: rc
     end function pigs_delta_action_parts

     ! replaces `Path(k,ip,ib) = xnew(k)` on accept
     function pigs_commit_beads(ctx,n,walker,ip,ib,x) bind(C,name='pigs_commit_beads') result(rc)
       import :: c_int, c_int32_t, c_int64_t, c_double, c_ptr
       type(c_ptr), value             :: ctx
       integer(c_int64_t), value      :: n
       integer(c_int32_t), intent(in) :: walker(*),ip(*),ib(*)
       real(c_double), intent(in)     :: x(*)
       integer(c_int) :: rc
     end function pigs_commit_beads

     function pigs_swap_tails(ctx,walker,iw,ik) bind(C,name='pigs_swap_tails') result(rc)
       import :: c_int, c_int32_t, c_ptr
       type(c_ptr), value        :: ctx
       integer(c_int32_t), value :: walker,iw,ik
       integer(c_int) :: rc
     end function pigs_swap_tails

     ! PotentialEnergy (reference sample_mod.f90:13)
     function pigs_potential_energy_slice(ctx,walker,ib,want_F2,Pot,F2) &
          & bind(C,name='pigs_potential_energy_slice') result(rc)
       import :: c_int, c_int32_t, c_double, c_ptr
       type(c_ptr), value        :: ctx
       integer(c_int32_t), value :: walker,ib,want_F2
       real(c_double)            :: Pot,F2
       integer(c_int) :: rc
     end function pigs_potential_energy_slice

     ! ThermEnergy (reference sample_mod.f90:323)
     function pigs_therm_energy_batch(ctx,n,walkers,E,Ec,Ep) bind(C,name='pigs_therm_energy_batch') result(rc)
       import :: c_int, c_int32_t, c_double, c_ptr
       type(c_ptr), value             :: ctx
       integer(c_int32_t), value      :: n
       integer(c_int32_t), intent(in) :: walkers(*)
       real(c_double)                 :: E(*),Ec(*),Ep(*)
       integer(c_int) :: rc
     end function pigs_therm_energy_batch

     ! LocalEnergy (reference sample_mod.f90:154)
     function pigs_local_energy_batch(ctx,n,walkers,ib,E,Kin,Pot) bind(C,name='pigs_local_energy_batch') result(rc)
       import :: c_int, c_int32_t, c_double, c_ptr
       type(c_ptr), value             :: ctx
       integer(c_int32_t), value      :: n,ib
       integer(c_int32_t), intent(in) :: walkers(*)
       real(c_double)                 :: E(*),Kin(*),Pot(*)
       integer(c_int) :: rc
     end function pigs_local_energy_batch

     ! K6: device-resident sampler (diagonal sector, sampling='bis')
     function pigs_sampler_init(ctx,sp) bind(C,name='pigs_sampler_init') result(rc)
       import :: c_int, c_ptr, pigs_sweep_params
       type(c_ptr), value :: ctx
       type(pigs_sweep_params), intent(in) :: sp
       integer(c_int) :: rc
     end function pigs_sampler_init

     function pigs_sampler_set_rng(ctx,walker,mti,mt) bind(C,name='pigs_sampler_set_rng') result(rc)
       import :: c_int, c_int32_t, c_ptr
       type(c_ptr), value             :: ctx
       integer(c_int32_t), value      :: walker,mti
       integer(c_int32_t), intent(in) :: mt(0:623)
       integer(c_int) :: rc
     end function pigs_sampler_set_rng

     function pigs_sampler_get_rng(ctx,walker,mti,mt) bind(C,name='pigs_sampler_get_rng') result(rc)
       import :: c_int, c_int32_t, c_ptr
       type(c_ptr), value        :: ctx
       integer(c_int32_t), value :: walker
       integer(c_int32_t)        :: mti,mt(0:623)
       integer(c_int) :: rc
     end function pigs_sampler_get_rng

     function pigs_sampler_step(ctx,istep) bind(C,name='pigs_sampler_step') result(rc)
       import :: c_int, c_int32_t, c_ptr
       type(c_ptr), value        :: ctx
       integer(c_int32_t), value :: istep
       integer(c_int) :: rc
     end function pigs_sampler_step

     function pigs_sampler_counters(ctx,acc) bind(C,name='pigs_sampler_counters') result(rc)
       import :: c_int, c_int64_t, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int64_t) :: acc(*)
       integer(c_int) :: rc
     end function pigs_sampler_counters

     function pigs_sampler_counters16(ctx,cnt) bind(C,name='pigs_sampler_counters16') result(rc)
       import :: c_int, c_int64_t, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int64_t) :: cnt(*)
       integer(c_int) :: rc
     end function pigs_sampler_counters16

     function pigs_sampler_get_worm(ctx,isopen,iworm,xend) bind(C,name='pigs_sampler_get_worm') result(rc)
       import :: c_int, c_int32_t, c_double, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int32_t) :: isopen(*),iworm(*)
       real(c_double)     :: xend(*)
       integer(c_int) :: rc
     end function pigs_sampler_get_worm

     function pigs_sampler_set_worm(ctx,isopen,iworm,xend) bind(C,name='pigs_sampler_set_worm') result(rc)
       import :: c_int, c_int32_t, c_double, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int32_t), intent(in) :: isopen(*),iworm(*)
       real(c_double), intent(in)     :: xend(*)
       integer(c_int) :: rc
     end function pigs_sampler_set_worm

     function pigs_sampler_events(ctx,events) bind(C,name='pigs_sampler_events') result(rc)
       import :: c_int, c_int32_t, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int32_t) :: events(*)
       integer(c_int) :: rc
     end function pigs_sampler_events

     function pigs_sampler_nrho(ctx,nrho,reset) bind(C,name='pigs_sampler_nrho') result(rc)
       import :: c_int, c_int32_t, c_double, c_ptr
       type(c_ptr), value             :: ctx
       real(c_double)                 :: nrho(*)
       integer(c_int32_t), intent(in) :: reset(*)      ! per walker: 1 = zero the histogram after the copy
       integer(c_int) :: rc
     end function pigs_sampler_nrho

     function pigs_slice_download(ctx,ib,R) bind(C,name='pigs_slice_download') result(rc)
       import :: c_int, c_int32_t, c_double, c_ptr
       type(c_ptr), value        :: ctx
       integer(c_int32_t), value :: ib
       real(c_double)            :: R(*)
       integer(c_int) :: rc
     end function pigs_slice_download

     ! K7: PairCorrelation + StructureFactor increments of slice ib (reference sample_mod.f90:392-473)
     function pigs_structure_batch(ctx,n,walkers,ib,Nbin,rbin,Nk,gr,Sk) bind(C,name='pigs_structure_batch') result(rc)
       import :: c_int, c_int32_t, c_double, c_ptr
       type(c_ptr), value             :: ctx
       integer(c_int32_t), value      :: n,ib,Nbin,Nk
       integer(c_int32_t), intent(in) :: walkers(*)
       real(c_double), value          :: rbin
       real(c_double)                 :: gr(*),Sk(*)
       integer(c_int) :: rc
     end function pigs_structure_batch

     ! every estimator of a diagonal MC step in one call (reference vpi.f90:443-469): en(1:3,i) = LocalEnergy at slice 0,
     ! en(4:6,i) at slice 2Nb, en(7:9,i) = ThermEnergy of walker walkers(i); gr / Sk = c_null_ptr: no structural estimators
     function pigs_diagonal_estimators(ctx,n,walkers,Nbin,rbin,Nk,en,gr,Sk) bind(C,name='pigs_diagonal_estimators') result(rc)
       import :: c_int, c_int32_t, c_double, c_ptr
       type(c_ptr), value             :: ctx
       integer(c_int32_t), value      :: n,Nbin,Nk
       integer(c_int32_t), intent(in) :: walkers(*)
       real(c_double), value          :: rbin
       real(c_double)                 :: en(9,*)
       type(c_ptr), value             :: gr,Sk
       integer(c_int) :: rc
     end function pigs_diagonal_estimators

     ! the same overlapped with the next step of the device-resident sampler: _begin snapshots the worldlines and queues the
     ! estimator kernels on the context's second stream, _end collects (layouts as pigs_diagonal_estimators)
     function pigs_diagonal_estimators_begin(ctx,n,walkers,Nbin,rbin,Nk,structure) &
          & bind(C,name='pigs_diagonal_estimators_begin') result(rc)
       import :: c_int, c_int32_t, c_double, c_ptr
       type(c_ptr), value             :: ctx
       integer(c_int32_t), value      :: n,Nbin,Nk,structure
       integer(c_int32_t), intent(in) :: walkers(*)
       real(c_double), value          :: rbin
       integer(c_int) :: rc
     end function pigs_diagonal_estimators_begin

     function pigs_diagonal_estimators_end(ctx,en,gr,Sk) bind(C,name='pigs_diagonal_estimators_end') result(rc)
       import :: c_int, c_double, c_ptr
       type(c_ptr), value :: ctx
       real(c_double)     :: en(9,*)
       type(c_ptr), value :: gr,Sk
       integer(c_int) :: rc
     end function pigs_diagonal_estimators_end

     function pigs_sampler_event_ints(ctx,n) bind(C,name='pigs_sampler_event_ints') result(rc)
       import :: c_int, c_int32_t, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int32_t) :: n
       integer(c_int) :: rc
     end function pigs_sampler_event_ints

     function pigs_comm_init_all(ctxs,nranks) bind(C,name='pigs_comm_init_all') result(rc)
       import :: c_int, c_int32_t, c_ptr
       type(c_ptr)               :: ctxs(*)
       integer(c_int32_t), value :: nranks
       integer(c_int) :: rc
     end function pigs_comm_init_all

     function pigs_estimators_allreduce(ctx,vec,n) bind(C,name='pigs_estimators_allreduce') result(rc)
       import :: c_int, c_int32_t, c_double, c_ptr
       type(c_ptr), value        :: ctx
       real(c_double)            :: vec(*)
       integer(c_int32_t), value :: n
       integer(c_int) :: rc
     end function pigs_estimators_allreduce

  end interface

contains

  ! Stop with the library's error text: the host-side policy (the library itself never stops).
  subroutine pigs_check(rc,what)
    integer(c_int), intent(in)   :: rc
    character(len=*), intent(in) :: what
    character(kind=c_char), pointer :: s(:)
    type(c_ptr) :: p
    integer :: n
    if (rc==PIGS_OK) return
    p = pigs_last_error()
    write (0,'(a,a,a,i0)') 'pigs: ',what,' failed with status ',rc
    if (c_associated(p)) then
       call c_f_pointer(p,s,[512])
       n = 1
       do while (n<512 .and. s(n)/=c_null_char)
          n = n+1
       end do
       write (0,'(512a1)') s(1:n-1)
    end if
    stop 1
  end subroutine pigs_check

end module pigs_capi
