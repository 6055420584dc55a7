! ref_probe.f90 -- TEST INFRASTRUCTURE (oracle side).  NOT part of the product.
!
! bind(C) wrappers around the *unmodified* reference routines of
! amaciarey/PathIntegralGroundState so that tests can call the reference
! directly (ctypes) and dump full 64-bit outputs.  The reference sources are
! compiled where they lie under /root/reference by oracle/Makefile; only this
! probe (our own code) lives in the repo.  Outputs go to oracle/_ref/.
!
! Each wrapper names the reference procedure it forwards to (file:line).

module ref_probe

  use iso_c_binding
  use random_mod      ! /root/reference/random_mod.f90
  use global_mod      ! /root/reference/global_mod.f90
  use system_mod      ! /root/reference/system_mod.f90
  use pbc_mod         ! /root/reference/pbc_mod.f90
  use vpi_mod         ! /root/reference/vpi_mod.f90
  use sample_mod      ! /root/reference/sample_mod.f90

  implicit none

contains

  !---------------------------------------------------------------------
  ! Module globals the hot path reads implicitly (global_mod.f90:5-12,
  ! system_mod.f90:8-9); set the way vpi.f90:80-128 sets them.
  subroutine ref_set_globals(c_dim,c_np,c_nb,c_nmax,c_lbox,c_rcut,c_rm,c_aho,&
       & c_wf_table,c_v_table,c_nbin,c_npw,c_cworm) bind(C,name='ref_set_globals')
    integer(c_int), value :: c_dim,c_np,c_nb,c_nmax,c_wf_table,c_v_table,c_nbin,c_npw
    real(c_double), value :: c_rcut,c_rm,c_cworm
    real(c_double)        :: c_lbox(c_dim),c_aho(c_dim)
    integer :: k

    dim   = c_dim
    Np    = c_np
    Nb    = c_nb
    Nmax  = c_nmax
    Nbin  = c_nbin
    Npw   = c_npw
    CWorm = c_cworm
    pi    = acos(-1.d0)

    if (allocated(Lbox))     deallocate(Lbox)
    if (allocated(LboxHalf)) deallocate(LboxHalf)
    if (allocated(qbin))     deallocate(qbin)
    if (allocated(a_ho))     deallocate(a_ho)
    allocate(Lbox(dim),LboxHalf(dim),qbin(dim),a_ho(dim))

    do k=1,dim
       Lbox(k)     = c_lbox(k)
       LboxHalf(k) = 0.5d0*Lbox(k)          ! vpi.f90:118
       qbin(k)     = 2.d0*pi/Lbox(k)        ! vpi.f90:119
       a_ho(k)     = c_aho(k)
    end do

    rcut     = c_rcut
    rcut2    = rcut*rcut                    ! vpi.f90:127
    rbin     = rcut/real(Nbin)              ! vpi.f90:128
    Rm       = c_rm
    wf_table = (c_wf_table/=0)
    v_table  = (c_v_table/=0)

  end subroutine ref_set_globals

  ! Box length exactly as vpi.f90:112 computes it (single-precision real()).
  function ref_box_length(c_np,c_dim,density) bind(C,name='ref_box_length') result(L)
    integer(c_int), value :: c_np,c_dim
    real(c_double), value :: density
    real(c_double) :: L
    L = (real(c_np)/density)**(1.d0/real(c_dim))
  end function ref_box_length

  subroutine ref_set_dr(c_dr) bind(C,name='ref_set_dr')
    real(c_double), value :: c_dr
    dr = c_dr
  end subroutine ref_set_dr

  function ref_get_dr() bind(C,name='ref_get_dr') result(x)
    real(c_double) :: x
    x = dr
  end function ref_get_dr

  !---------------------------------------------------------------------
  ! vpi_mod.f90:84-145 (these also write jastrow.out / potential.out in cwd)
  subroutine ref_jastrow_table(rmax,WF) bind(C,name='ref_jastrow_table')
    real(c_double), value :: rmax
    real(c_double) :: WF(0:Nmax+1)
    call JastrowTable(rmax,WF)
  end subroutine ref_jastrow_table

  subroutine ref_potential_table(rmax,VT) bind(C,name='ref_potential_table')
    real(c_double), value :: rmax
    real(c_double) :: VT(0:Nmax+1)
    call PotentialTable(rmax,VT)
  end subroutine ref_potential_table

  ! system_mod.f90:136-182 / 38-66
  function ref_potential(rij) bind(C,name='ref_potential') result(v)
    real(c_double), value :: rij
    real(c_double) :: v
    real(kind=8) :: x(dim)
    x = 0.d0
    v = Potential(x,rij)
  end function ref_potential

  function ref_logpsi(opt,c_rm,rij) bind(C,name='ref_logpsi') result(v)
    integer(c_int), value :: opt
    real(c_double), value :: c_rm,rij
    real(c_double) :: v
    v = LogPsi(opt,c_rm,rij)
  end function ref_logpsi

  !---------------------------------------------------------------------
  ! Numeric primitives
  function ref_interpolate(opt,N,dx,F,x) bind(C,name='ref_interpolate') result(v)
    integer(c_int), value :: opt,N
    real(c_double), value :: dx,x
    real(c_double) :: F(0:N+1)
    real(c_double) :: v
    real(kind=8)   :: Interpolate            ! interpolate.f90:1
    v = Interpolate(opt,N,dx,F,x)
  end function ref_interpolate

  subroutine ref_minimum_image(xij,rij2) bind(C,name='ref_minimum_image')
    real(c_double) :: xij(dim),rij2
    call MinimumImage(xij,rij2)              ! pbc_mod.f90:29
  end subroutine ref_minimum_image

  function ref_green_function(opt,ib,dt,Pot,F2) bind(C,name='ref_green_function') result(v)
    integer(c_int), value :: opt,ib
    real(c_double), value :: dt,Pot,F2
    real(c_double) :: v
    integer :: o,i
    real(kind=8) :: d,p,f
    o = opt; i = ib; d = dt; p = Pot; f = F2
    v = GreenFunction(o,i,d,p,f)             ! global_mod.f90:19
  end function ref_green_function

  !---------------------------------------------------------------------
  ! Hot path, sampling side (vpi_mod.f90:2491-2841).  ip is 1-based, ib 0-based
  ! exactly as in the reference.
  subroutine ref_update_action(c_trap,LogWF,VT,Path,c_ip,c_ib,xnew,xold,c_dt,DeltaS) &
       & bind(C,name='ref_update_action')
    integer(c_int), value :: c_trap,c_ip,c_ib
    real(c_double), value :: c_dt
    real(c_double) :: LogWF(0:Nmax+1),VT(0:Nmax+1)
    real(c_double) :: Path(dim,Np,0:2*Nb),xnew(dim),xold(dim),DeltaS
    logical :: trap
    integer :: ip,ib
    real(kind=8) :: dt
    trap = (c_trap/=0); ip = c_ip; ib = c_ib; dt = c_dt
    call UpdateAction(trap,LogWF,VT,Path,ip,ib,xnew,xold,dt,DeltaS)
  end subroutine ref_update_action

  subroutine ref_update_pot(c_trap,VT,c_ip,R,xnew,xold,DeltaPot,DeltaF2,want_f2) &
       & bind(C,name='ref_update_pot')
    integer(c_int), value :: c_trap,c_ip,want_f2
    real(c_double) :: VT(0:Nmax+1),R(dim,Np),xnew(dim),xold(dim),DeltaPot,DeltaF2
    logical :: trap
    integer :: ip
    trap = (c_trap/=0); ip = c_ip
    if (want_f2/=0) then
       call UpdatePot(trap,VT,ip,R,xnew,xold,DeltaPot,DeltaF2)
    else
       call UpdatePot(trap,VT,ip,R,xnew,xold,DeltaPot)
       DeltaF2 = 0.d0
    end if
  end subroutine ref_update_pot

  subroutine ref_update_wf(c_trap,LogWF,c_ip,R,xnew,xold,DeltaPsi) bind(C,name='ref_update_wf')
    integer(c_int), value :: c_trap,c_ip
    real(c_double) :: LogWF(0:Nmax+1),R(dim,Np),xnew(dim),xold(dim),DeltaPsi
    logical :: trap
    integer :: ip
    trap = (c_trap/=0); ip = c_ip
    call UpdateWf(trap,LogWF,ip,R,xnew,xold,DeltaPsi)
  end subroutine ref_update_wf

  !---------------------------------------------------------------------
  ! Hot path, estimator side (sample_mod.f90:13-388)
  subroutine ref_potential_energy(c_trap,VT,R,Pot,F2,want_f2) bind(C,name='ref_potential_energy')
    integer(c_int), value :: c_trap,want_f2
    real(c_double) :: VT(0:Nmax+1),R(dim,Np),Pot,F2
    logical :: trap
    trap = (c_trap/=0)
    if (want_f2/=0) then
       call PotentialEnergy(trap,VT,R,Pot,F2)
    else
       call PotentialEnergy(trap,VT,R,Pot)
       F2 = 0.d0
    end if
  end subroutine ref_potential_energy

  subroutine ref_local_energy(c_trap,LogWF,VT,R,E,Kin,Pot) bind(C,name='ref_local_energy')
    integer(c_int), value :: c_trap
    real(c_double) :: LogWF(0:Nmax+1),VT(0:Nmax+1),R(dim,Np),E,Kin,Pot
    logical :: trap
    trap = (c_trap/=0)
    call LocalEnergy(trap,LogWF,VT,R,E,Kin,Pot)
  end subroutine ref_local_energy

  subroutine ref_therm_energy(c_trap,VT,Path,c_dt,E,Ec,Ep) bind(C,name='ref_therm_energy')
    integer(c_int), value :: c_trap
    real(c_double), value :: c_dt
    real(c_double) :: VT(0:Nmax+1),Path(dim,Np,0:2*Nb),E,Ec,Ep
    logical :: trap
    real(kind=8) :: dt
    trap = (c_trap/=0); dt = c_dt
    call ThermEnergy(trap,VT,Path,dt,E,Ec,Ep)
  end subroutine ref_therm_energy

  subroutine ref_pair_correlation(R,gr) bind(C,name='ref_pair_correlation')
    real(c_double) :: R(dim,Np),gr(Nbin)
    call PairCorrelation(R,gr)               ! sample_mod.f90:392
  end subroutine ref_pair_correlation

  subroutine ref_structure_factor(c_nk,R,Sk) bind(C,name='ref_structure_factor')
    integer(c_int), value :: c_nk
    real(c_double) :: R(dim,Np),Sk(dim,c_nk)
    integer :: Nk
    Nk = c_nk
    call StructureFactor(Nk,R,Sk)            ! sample_mod.f90:432
  end subroutine ref_structure_factor

  subroutine ref_obdm(xend,nrho) bind(C,name='ref_obdm')
    real(c_double) :: xend(dim,2),nrho(0:Npw,Nbin)
    call OBDM(xend,nrho)                     ! sample_mod.f90:477
  end subroutine ref_obdm

  !---------------------------------------------------------------------
  ! RNG (random_mod.f90) and initial configuration (vpi_mod.f90:149-259)
  subroutine ref_sgrnd(seed) bind(C,name='ref_sgrnd')
    integer(c_int), value :: seed
    integer :: s
    s = seed
    call sgrnd(s)
  end subroutine ref_sgrnd

  function ref_grnd() bind(C,name='ref_grnd') result(x)
    real(c_double) :: x
    x = grnd()
  end function ref_grnd

  subroutine ref_rangauss(sigma,mu,x1,x2) bind(C,name='ref_rangauss')
    real(c_double), value :: sigma,mu
    real(c_double) :: x1,x2
    real(kind=8) :: s,m
    s = sigma; m = mu
    call rangauss(s,m,x1,x2)
  end subroutine ref_rangauss

  ! RNG state access (COMMON /block/ of random_mod.f90:17-18)
  subroutine ref_rng_get_state(c_mti,c_mt) bind(C,name='ref_rng_get_state')
    integer(c_int) :: c_mti,c_mt(0:623)
    integer :: mti,mt(0:623)
    common /block/mti,mt
    c_mti = mti
    c_mt  = mt
  end subroutine ref_rng_get_state

  subroutine ref_rng_set_state(c_mti,c_mt) bind(C,name='ref_rng_set_state')
    integer(c_int), value :: c_mti
    integer(c_int) :: c_mt(0:623)
    integer :: mti,mt(0:623)
    common /block/mti,mt
    mti = c_mti
    mt  = c_mt
  end subroutine ref_rng_set_state

  subroutine ref_init(c_trap,seed,Path,xend) bind(C,name='ref_init')
    integer(c_int), value :: c_trap,seed
    real(c_double) :: Path(dim,Np,0:2*Nb),xend(dim,2)
    logical :: trap,crystal,resume,isopen
    integer :: s,iworm
    trap = (c_trap/=0); crystal = .false.; resume = .false.
    isopen = .false.; iworm = 0; s = seed
    call init(trap,s,Path,xend,crystal,resume,isopen,iworm)
  end subroutine ref_init

  !---------------------------------------------------------------------
  ! Move set (vpi_mod.f90:313-2487): the callers of the hot path.  Exposed so
  ! that the host restatement can be compared move by move on identical RNG
  ! state.  `acc` is the reference's running acceptance counter.
  subroutine ref_translate_chain(c_trap,delta,LogWF,VT,c_dt,c_ip,Path,acc) &
       & bind(C,name='ref_translate_chain')
    integer(c_int), value :: c_trap,c_ip
    real(c_double), value :: delta,c_dt
    real(c_double) :: LogWF(0:Nmax+1),VT(0:Nmax+1),Path(dim,Np,0:2*Nb)
    integer(c_int) :: acc
    logical :: trap
    integer :: ip,a
    real(kind=8) :: d,dt
    trap = (c_trap/=0); ip = c_ip; a = acc; d = delta; dt = c_dt
    call TranslateChain(trap,d,LogWF,VT,dt,ip,Path,a)
    acc = a
  end subroutine ref_translate_chain

  ! which: 1 Staging, 2 MoveHead, 3 MoveTail, 4 Bisection, 5 MoveHeadBisection,
  !        6 MoveTailBisection.  par = Lstag (1-3) or Nlev (4-6).
  subroutine ref_diag_move(which,c_trap,LogWF,VT,c_dt,par,c_ip,Path,acc) &
       & bind(C,name='ref_diag_move')
    integer(c_int), value :: which,c_trap,par,c_ip
    real(c_double), value :: c_dt
    real(c_double) :: LogWF(0:Nmax+1),VT(0:Nmax+1),Path(dim,Np,0:2*Nb)
    integer(c_int) :: acc
    logical :: trap
    integer :: ip,a,p
    real(kind=8) :: dt
    trap = (c_trap/=0); ip = c_ip; a = acc; p = par; dt = c_dt
    select case (which)
    case (1); call Staging(trap,LogWF,VT,dt,p,ip,Path,a)
    case (2); call MoveHead(trap,LogWF,VT,dt,p,ip,Path,a)
    case (3); call MoveTail(trap,LogWF,VT,dt,p,ip,Path,a)
    case (4); call Bisection(trap,LogWF,VT,dt,p,ip,Path,a)
    case (5); call MoveHeadBisection(trap,LogWF,VT,dt,p,ip,Path,a)
    case (6); call MoveTailBisection(trap,LogWF,VT,dt,p,ip,Path,a)
    end select
    acc = a
  end subroutine ref_diag_move

  ! which: 1 TranslateHalfChain, 2 StagingHalfChain, 3 MoveHeadHalfChain,
  !        4 MoveTailHalfChain.
  subroutine ref_half_move(which,c_trap,half,delta,LogWF,VT,c_dt,Lstag,c_ip,Path,xend,acc) &
       & bind(C,name='ref_half_move')
    integer(c_int), value :: which,c_trap,half,Lstag,c_ip
    real(c_double), value :: delta,c_dt
    real(c_double) :: LogWF(0:Nmax+1),VT(0:Nmax+1),Path(dim,Np,0:2*Nb),xend(dim,2)
    integer(c_int) :: acc
    logical :: trap
    integer :: ip,a,h,L
    real(kind=8) :: d,dt
    trap = (c_trap/=0); ip = c_ip; a = acc; h = half; L = Lstag; d = delta; dt = c_dt
    select case (which)
    case (1); call TranslateHalfChain(trap,h,d,LogWF,VT,dt,ip,Path,xend,a)
    case (2); call StagingHalfChain(trap,h,LogWF,VT,dt,L,ip,Path,xend,a)
    case (3); call MoveHeadHalfChain(trap,h,LogWF,VT,dt,L,ip,Path,xend,a)
    case (4); call MoveTailHalfChain(trap,h,LogWF,VT,dt,L,ip,Path,xend,a)
    end select
    acc = a
  end subroutine ref_half_move

  subroutine ref_open_chain(c_trap,LogWF,VT,density,c_dt,Lstag,c_ip,Path,xend,c_isopen,acc) &
       & bind(C,name='ref_open_chain')
    integer(c_int), value :: c_trap,Lstag,c_ip
    real(c_double), value :: density,c_dt
    real(c_double) :: LogWF(0:Nmax+1),VT(0:Nmax+1),Path(dim,Np,0:2*Nb),xend(dim,2)
    integer(c_int) :: c_isopen,acc
    logical :: trap,isopen,newc
    integer :: ip,a,L
    real(kind=8) :: dens,dt
    trap = (c_trap/=0); ip = c_ip; a = acc; L = Lstag; dens = density; dt = c_dt
    isopen = (c_isopen/=0); newc = .false.
    call OpenChain(trap,LogWF,VT,dens,dt,L,ip,Path,xend,isopen,a,newc)
    acc = a
    c_isopen = merge(1,0,isopen)
  end subroutine ref_open_chain

  subroutine ref_close_chain(c_trap,LogWF,VT,density,c_dt,Lstag,c_ip,Path,xend,c_isopen,acc) &
       & bind(C,name='ref_close_chain')
    integer(c_int), value :: c_trap,Lstag,c_ip
    real(c_double), value :: density,c_dt
    real(c_double) :: LogWF(0:Nmax+1),VT(0:Nmax+1),Path(dim,Np,0:2*Nb),xend(dim,2)
    integer(c_int) :: c_isopen,acc
    logical :: trap,isopen,endc
    integer :: ip,a,L
    real(kind=8) :: dens,dt
    trap = (c_trap/=0); ip = c_ip; a = acc; L = Lstag; dens = density; dt = c_dt
    isopen = (c_isopen/=0); endc = .false.
    call CloseChain(trap,LogWF,VT,dens,dt,L,ip,Path,xend,isopen,a,endc)
    acc = a
    c_isopen = merge(1,0,isopen)
  end subroutine ref_close_chain

  subroutine ref_swap(c_trap,LogWF,VT,c_dt,Lstag,c_iw,Path,xend,acc,c_ik,c_swapped) &
       & bind(C,name='ref_swap')
    integer(c_int), value :: c_trap,Lstag
    real(c_double), value :: c_dt
    real(c_double) :: LogWF(0:Nmax+1),VT(0:Nmax+1),Path(dim,Np,0:2*Nb),xend(dim,2)
    integer(c_int) :: c_iw,acc,c_ik,c_swapped
    logical :: trap,swapped
    integer :: iw,a,L,ik
    real(kind=8) :: dt
    trap = (c_trap/=0); iw = c_iw; a = acc; L = Lstag; dt = c_dt
    ik = 0; swapped = .false.
    call Swap(trap,LogWF,VT,dt,L,iw,Path,xend,a,ik,swapped)
    acc = a; c_iw = iw; c_ik = ik
    c_swapped = merge(1,0,swapped)
  end subroutine ref_swap

end module ref_probe
