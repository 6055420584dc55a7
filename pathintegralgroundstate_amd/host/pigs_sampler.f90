!-----------------------------------------------------------------------
! pigs_sampler -- host side of the boundary: W independent PIGS walkers advanced in
! lock-step, every Delta-S evaluated on the GPU in one batch per stage.
!
! What the reference does one walker / one bead at a time (move set, reference
! vpi_mod.f90:313-2487: 24 `call UpdateAction` sites) is restated here as
!   (A) per walker: draw the proposal of the current move from the walker's own
!       MT19937 stream, write it into the host mirror of the worldline, queue one
!       item (walker, ip, ib, xnew, xold) per displaced bead;
!   (B) ONE call pigs_delta_action_batch for all walkers' items;
!   (C) per walker: sum its Delta S in the reference's order, Metropolis test with
!       the walker's next uniform, then commit (device) or restore (host mirror).
! Legal because every Delta S of one proposal depends only on OTHER particles'
! coordinates at the same slice, which do not change during the proposal.  Multi-level
! moves (bisection) run (A)-(C) once per level; walkers rejected at a level drop out, so
! each walker consumes exactly the random numbers the reference consumes for its seed.
!
! Arithmetic follows the reference expression by expression, including its
! single-precision real() conversions (SURVEY.md quirk Q7), so that trajectories are
! decision-exact for identical seeds.
!-----------------------------------------------------------------------
module pigs_sampler

  use iso_c_binding
  use pigs_capi
  use pigs_rng

  implicit none
  private
  public :: sampler_t, sampler_init, sampler_free, sampler_upload, sampler_flush
  public :: mv_translate, mv_bisection, mv_end_bisection, mv_staging, mv_end_staging
  public :: mv_translate_half, mv_staging_half, mv_end_staging_half
  public :: mv_open, mv_close, mv_swap
  public :: HEAD, TAIL

  integer, parameter :: HEAD = 1, TAIL = 2

  type sampler_t
     ! ---- system (the reference's module globals)
     integer :: dim = 3, Np = 0, Nb = 0, W = 0
     logical :: trap = .false.
     real(8) :: dt = 0.d0, density = 0.d0, CWorm = 0.d0, pi = 0.d0
     real(8) :: Lbox(3) = 1.d0, LboxHalf(3) = 0.5d0
     ! ---- state of every walker
     real(8), allocatable :: Path(:,:,:,:)      ! (dim,Np,0:2*Nb,W): host mirror of the resident worldlines
     real(8), allocatable :: xend(:,:,:)        ! (dim,2,W): the two worm ends at bead Nb
     logical, allocatable :: isopen(:)
     integer, allocatable :: iworm(:)
     type(mt_state), allocatable :: rng(:)
     type(c_ptr) :: ctx = c_null_ptr
     ! ---- proposal items of the current stage (SoA, as the C ABI takes them)
     integer :: n_items = 0, cap = 0
     ! (views of the library's pinned, device-mapped staging arrays: the kernel reads them in place)
     integer(c_int32_t), pointer :: it_w(:) => null(), it_ip(:) => null(), it_ib(:) => null()
     real(c_double), pointer     :: it_xnew(:,:) => null(), it_xold(:,:) => null(), it_dS(:) => null()
     real(c_double), allocatable :: it_wgt(:)
     ! ---- beads to write to the device before the next evaluation
     integer :: n_commit = 0, ccap = 0
     ! pigs_commit_staged is asynchronous and its kernel reads the pinned arrays below IN PLACE: after a flush they may not
     ! be written until a synchronising call has passed (round 3: a Swap stage in which no walker got as far as a proposal
     ! flushed and returned without one, the next mover's select_half overwrote slot 1 while the kernel was still on its
     ! way, and one accepted bead never reached the device -- found by scripts/k6_vs_host_soak.py, ~1 in 100 MC steps)
     logical :: commit_in_flight = .false.
     integer(c_int32_t), pointer :: cm_w(:) => null(), cm_ip(:) => null(), cm_ib(:) => null()
     real(c_double), pointer     :: cm_x(:,:) => null()
     ! ---- per-walker bookkeeping of the move in flight
     integer, allocatable :: first(:), cnt(:), want(:), seg_i(:), seg_e(:), aux_i(:), aux_k(:)
     logical, allocatable :: alive(:), flag(:)
     real(8), allocatable :: S0(:), DK(:)
     real(8), allocatable :: Old(:,:,:), Old2(:,:,:)  ! (dim,0:2*Nb,W) saved chains
     ! ---- statistics
     integer(8) :: n_eval_items = 0, n_eval_calls = 0
     real(8)    :: t_eval = 0.d0         ! wall seconds spent inside the batched Delta-S calls
     ! ---- host threads for the per-walker loops of a stage (proposals, decisions): walkers are independent there --
     ! own random stream, own item slots (plan_items), own rows of the mirror -- so the threads change no result
     integer    :: nthr = 1
  end type sampler_t

contains

  !=====================================================================
  ! set-up
  !=====================================================================
  subroutine sampler_init(s,dim,Np,Nb,W,trap,dt,density,CWorm,Lbox,ctx)
    type(sampler_t), intent(inout) :: s
    integer, intent(in) :: dim,Np,Nb,W
    logical, intent(in) :: trap
    real(8), intent(in) :: dt,density,CWorm,Lbox(dim)
    type(c_ptr), intent(in) :: ctx
    integer :: k
    s%dim = dim; s%Np = Np; s%Nb = Nb; s%W = W; s%trap = trap
    s%dt = dt; s%density = density; s%CWorm = CWorm; s%ctx = ctx
    s%pi = acos(-1.d0)
    s%nthr = host_threads(W)
    s%n_items = 0; s%cap = 0; s%n_commit = 0; s%ccap = 0; s%commit_in_flight = .false.
    s%n_eval_items = 0; s%n_eval_calls = 0; s%t_eval = 0.d0
    do k=1,dim
       s%Lbox(k) = Lbox(k)
       s%LboxHalf(k) = 0.5d0*Lbox(k)
    end do
    allocate(s%Path(dim,Np,0:2*Nb,W), s%xend(dim,2,W), s%isopen(W), s%iworm(W), s%rng(W))
    allocate(s%first(W), s%cnt(W), s%want(W), s%seg_i(W), s%seg_e(W), s%aux_i(W), s%aux_k(W))
    allocate(s%alive(W), s%flag(W), s%S0(W), s%DK(W))
    allocate(s%Old(dim,0:2*Nb,W), s%Old2(dim,0:2*Nb,W))
    s%isopen = .false.; s%iworm = 0; s%Path = 0.d0; s%xend = 0.d0
    call grow_items(s,max(64,W*(2*Nb+1)))
    call grow_commit(s,max(64,2*W*(2*Nb+1)))
  end subroutine sampler_init

  ! threads for the per-walker loops: PIGS_HOST_THREADS, else what OpenMP offers to this (possibly nested) level, at most
  ! 16 and at least 128 walkers per thread: a stage's host work is ~0.4 us per walker and a stage has four fork-joins of
  ! several us each -- measured at 128 walkers (N=256, 161 beads): 2.7 s per block on one thread, 3.0 s on eight
  function host_threads(W) result(n)
    !$ use omp_lib
    integer, intent(in) :: W
    integer :: n,ios
    character(len=32) :: ev
    n = 1
    !$ n = min(16,omp_get_max_threads())
    n = min(n,W/128)
    call get_environment_variable('PIGS_HOST_THREADS',ev)
    if (len_trim(ev)>0) then                                  ! explicit request: any count up to one thread per walker
       read(ev,*,iostat=ios) n
       if (ios/=0) n = 1
       n = min(n,W)
    end if
    n = max(1,n)
  end function host_threads

  subroutine sampler_free(s)
    type(sampler_t), intent(inout) :: s
    if (allocated(s%Path)) deallocate(s%Path,s%xend,s%isopen,s%iworm,s%rng,s%first,s%cnt,s%want, &
         & s%seg_i,s%seg_e,s%aux_i,s%aux_k,s%alive,s%flag,s%S0,s%DK,s%Old,s%Old2)
    if (allocated(s%it_wgt)) deallocate(s%it_wgt)
    nullify(s%it_w,s%it_ip,s%it_ib,s%it_xnew,s%it_xold,s%it_dS,s%cm_w,s%cm_ip,s%cm_ib,s%cm_x)   ! owned by the context
    s%cap = 0; s%ccap = 0
  end subroutine sampler_free

  ! push the host mirror of every walker to the device
  subroutine sampler_upload(s)
    type(sampler_t), intent(inout) :: s
    call pigs_check(pigs_path_upload_all(s%ctx,s%Path),'pigs_path_upload_all')
    s%n_commit = 0
    s%commit_in_flight = .false.
  end subroutine sampler_upload

  ! write every pending accepted bead to the device now (otherwise done lazily before the next
  ! evaluation): needed before anything else reads the resident worldlines (estimator kernels)
  subroutine sampler_flush(s)
    type(sampler_t), intent(inout) :: s
    call flush_commits(s)
    ! the staged commit is asynchronous: make sure it has consumed its arrays before they are reused
    call pigs_check(pigs_sync(s%ctx),'pigs_sync')
    s%commit_in_flight = .false.
  end subroutine sampler_flush

  subroutine grow_items(s,n)
    type(sampler_t), intent(inout) :: s
    integer, intent(in) :: n
    type(c_ptr) :: pw,pi,pb,pxn,pxo,pds
    real(c_double), allocatable :: g(:)
    integer :: m
    if (n<=s%cap) return
    m = max(n,2*s%cap)
    call pigs_check(pigs_stage_reserve(s%ctx,int(m,c_int64_t),int(s%n_items,c_int64_t),pw,pi,pb,pxn,pxo,pds), &
         & 'pigs_stage_reserve')
    call c_f_pointer(pw,s%it_w,[m]);  call c_f_pointer(pi,s%it_ip,[m]); call c_f_pointer(pb,s%it_ib,[m])
    call c_f_pointer(pxn,s%it_xnew,[s%dim,m]); call c_f_pointer(pxo,s%it_xold,[s%dim,m])
    call c_f_pointer(pds,s%it_dS,[m])
    allocate(g(m))
    if (s%n_items>0) g(1:s%n_items) = s%it_wgt(1:s%n_items)
    call move_alloc(g,s%it_wgt)
    s%cap = m
  end subroutine grow_items

  subroutine grow_commit(s,n)
    type(sampler_t), intent(inout) :: s
    integer, intent(in) :: n
    type(c_ptr) :: pw,pi,pb,px
    integer :: m
    if (n<=s%ccap) return
    m = max(n,2*s%ccap)
    call pigs_check(pigs_commit_reserve(s%ctx,int(m,c_int64_t),int(s%n_commit,c_int64_t),pw,pi,pb,px), &
         & 'pigs_commit_reserve')
    call c_f_pointer(pw,s%cm_w,[m]); call c_f_pointer(pi,s%cm_ip,[m]); call c_f_pointer(pb,s%cm_ib,[m])
    call c_f_pointer(px,s%cm_x,[s%dim,m])
    s%ccap = m
  end subroutine grow_commit

  !=====================================================================
  ! small pieces shared by every mover
  !=====================================================================

  ! single wrap of a coordinate into the box (reference pbc_mod.f90:11-25)
  subroutine wrap_coord(s,k,x)
    type(sampler_t), intent(in) :: s
    integer, intent(in)    :: k
    real(8), intent(inout) :: x
    if (s%trap) return
    if (x> s%LboxHalf(k)) x = x-s%Lbox(k)
    if (x<-s%LboxHalf(k)) x = x+s%Lbox(k)
  end subroutine wrap_coord

  ! nearest image of the anchor `a` seen from `xo`, on the "previous" side:
  ! xo + wrap(a - xo)   (e.g. reference vpi_mod.f90:517-522)
  function image_prev(s,k,a,xo) result(x)
    type(sampler_t), intent(in) :: s
    integer, intent(in) :: k
    real(8), intent(in) :: a,xo
    real(8) :: x
    x = a-xo
    if (.not. s%trap) then
       if (x<-s%LboxHalf(k)) x = x+s%Lbox(k)
       if (x> s%LboxHalf(k)) x = x-s%Lbox(k)
    end if
    x = xo+x
  end function image_prev

  ! same on the "next" side: xo - wrap(xo - a)   (e.g. vpi_mod.f90:524-529)
  function image_next(s,k,a,xo) result(x)
    type(sampler_t), intent(in) :: s
    integer, intent(in) :: k
    real(8), intent(in) :: a,xo
    real(8) :: x
    x = xo-a
    if (.not. s%trap) then
       if (x<-s%LboxHalf(k)) x = x+s%Lbox(k)
       if (x> s%LboxHalf(k)) x = x-s%Lbox(k)
    end if
    x = xo-x
  end function image_next

  ! squared nearest-image distance between two beads of one particle
  function link_r2(s,a,b) result(r2)
    type(sampler_t), intent(in) :: s
    real(8), intent(in) :: a(s%dim),b(s%dim)
    real(8) :: r2,x
    integer :: k
    r2 = 0.d0
    do k=1,s%dim
       x = a(k)-b(k)
       if (.not. s%trap) then
          if (x> s%LboxHalf(k)) x = x-s%Lbox(k)
          if (x<-s%LboxHalf(k)) x = x+s%Lbox(k)
       end if
       r2 = r2+x*x
    end do
  end function link_r2

  ! item slots of a stage are planned per walker (plan_items) so that walkers can generate their
  ! proposals concurrently: walker w owns slots first(w) .. first(w)+want(w)-1
  subroutine plan_items(s)
    type(sampler_t), intent(inout) :: s
    integer :: w,n
    n = 0
    do w=1,s%W
       s%first(w) = n+1
       s%cnt(w)   = 0
       n = n+s%want(w)
    end do
    call grow_items(s,n)
    s%n_items = n
  end subroutine plan_items

  subroutine add_item(s,w,ip,ib,xnew,xold,wgt)
    type(sampler_t), intent(inout) :: s
    integer, intent(in) :: w,ip,ib
    real(8), intent(in) :: xnew(s%dim),xold(s%dim),wgt
    integer :: n
    n = s%first(w)+s%cnt(w)
    s%it_w(n) = w-1; s%it_ip(n) = ip; s%it_ib(n) = ib
    s%it_xnew(:,n) = xnew; s%it_xold(:,n) = xold; s%it_wgt(n) = wgt
    s%cnt(w) = s%cnt(w)+1
  end subroutine add_item

  subroutine queue_commit(s,w,ip,ib,x)
    type(sampler_t), intent(inout) :: s
    integer, intent(in) :: w,ip,ib
    real(8), intent(in) :: x(s%dim)
    integer :: n
    if (s%commit_in_flight) then                 ! the previous flush may still be reading the staging arrays
       call pigs_check(pigs_sync(s%ctx),'pigs_sync')
       s%commit_in_flight = .false.
    end if
    if (s%n_commit+1>s%ccap) call grow_commit(s,s%n_commit+1)
    n = s%n_commit+1
    s%cm_w(n) = w-1; s%cm_ip(n) = ip; s%cm_ib(n) = ib; s%cm_x(:,n) = x
    s%n_commit = n
  end subroutine queue_commit

  ! beads ia..ie of particle ip (host mirror) -> device
  subroutine queue_commit_range(s,w,ip,ia,ie)
    type(sampler_t), intent(inout) :: s
    integer, intent(in) :: w,ip,ia,ie
    integer :: ib
    do ib=ia,ie
       call queue_commit(s,w,ip,ib,s%Path(:,ip,ib,w))
    end do
  end subroutine queue_commit_range

  subroutine flush_commits(s)
    type(sampler_t), intent(inout) :: s
    if (s%n_commit==0) return
    call pigs_check(pigs_commit_staged(s%ctx,int(s%n_commit,c_int64_t)),'pigs_commit_staged')
    s%n_commit = 0
    s%commit_in_flight = .true.
  end subroutine flush_commits

  subroutine begin_stage(s)
    type(sampler_t), intent(inout) :: s
    s%n_items = 0
    s%cnt = 0
    s%want = 0
    s%first = 1
  end subroutine begin_stage

  ! (B): every queued item through the GPU
  subroutine evaluate(s)
    type(sampler_t), intent(inout) :: s
    integer(8) :: c0,c1,rate
    call flush_commits(s)
    if (s%n_items==0) return
    call system_clock(c0,rate)
    call pigs_check(pigs_delta_action_staged(s%ctx,int(s%n_items,c_int64_t)),'pigs_delta_action_staged')
    s%commit_in_flight = .false.                 ! (synchronous: the commit kernel ahead of it in the stream is done)
    call system_clock(c1)
    s%t_eval = s%t_eval+dble(c1-c0)/dble(rate)
    s%n_eval_items = s%n_eval_items+s%n_items
    s%n_eval_calls = s%n_eval_calls+1
  end subroutine evaluate

  ! sum of the walker's weighted Delta S, in item order, starting from `s0`
  function sum_items(s,w,s0) result(t)
    type(sampler_t), intent(in) :: s
    integer, intent(in) :: w
    real(8), intent(in) :: s0
    real(8) :: t
    integer :: i
    t = s0
    do i=s%first(w),s%first(w)+s%cnt(w)-1
       if (s%it_wgt(i)==1.d0) then
          t = t+s%it_dS(i)
       else
          t = t+s%it_wgt(i)*s%it_dS(i)
       end if
    end do
  end function sum_items

  ! Metropolis question on exp(a) (reference e.g. vpi_mod.f90:356-364): no uniform is
  ! drawn when exp(a) >= 1
  function metropolis(s,w,a) result(acc)
    type(sampler_t), intent(inout) :: s
    integer, intent(in) :: w
    real(8), intent(in) :: a
    logical :: acc
    if (exp(a)>=1.d0) then
       acc = .true.
    else
       acc = exp(a)>=mt_real(s%rng(w))
    end if
  end function metropolis

  ! every mover takes its segment through save_chain before it touches a bead: the one place where a segment
  ! that leaves the chain is caught (this flang has no -fcheck=bounds and its -fsanitize=address does not
  ! instrument Fortran array accesses -- probed -- so the invariant is checked in the product build, always:
  ! two integer compares per move)
  subroutine need_beads(s,w,ip,ia,ie)
    type(sampler_t), intent(in) :: s
    integer, intent(in) :: w,ip,ia,ie
    if (ia<0 .or. ie>2*s%Nb .or. ia>ie .or. ip<1 .or. ip>s%Np .or. w<1 .or. w>s%W) then
       write (0,'(a,5i8,a,3i8)') ' pigs_sampler: segment outside the worldline (w,ip,ia,ie,2Nb): ',w,ip,ia,ie,2*s%Nb, &
            & '  (W,Np,Nb): ',s%W,s%Np,s%Nb
       error stop 3
    end if
  end subroutine need_beads

  subroutine save_chain(s,w,ip,ia,ie)
    type(sampler_t), intent(inout) :: s
    integer, intent(in) :: w,ip,ia,ie
    call need_beads(s,w,ip,ia,ie)
    s%Old(:,ia:ie,w) = s%Path(:,ip,ia:ie,w)
    s%seg_i(w) = ia; s%seg_e(w) = ie
  end subroutine save_chain

  subroutine restore_chain(s,w,ip)
    type(sampler_t), intent(inout) :: s
    integer, intent(in) :: w,ip
    s%Path(:,ip,s%seg_i(w):s%seg_e(w),w) = s%Old(:,s%seg_i(w):s%seg_e(w),w)
  end subroutine restore_chain

  ! A worm-sector segment ii..ie that leaves the chain 0..2Nb (only possible when Lstag > Nb).  The reference then
  ! indexes Path and OldChain outside their bounds (vpi_mod.f90:1853-1857 with ii = Nb-Ls < 0) and survives by
  ! luck; with CWorm = 0 the proposal is never accepted whatever it read (quirk Q11: SumDeltaS starts at +Inf or
  ! ends as NaN, exp() of it is never >= a uniform), so the only thing that reaches the rest of the run is the
  ! position of the random stream.  Such a proposal is therefore DRAWN, NOT BUILT: no load, no store, no item.
  logical function seg_outside(s,ii,ie)
    type(sampler_t), intent(in) :: s
    integer, intent(in) :: ii,ie
    seg_outside = (ii<0 .or. ie>2*s%Nb .or. ii>ie)
  end function seg_outside

  ! the random numbers of a proposal with `ng` Gaussian draws followed by a Metropolis question that a uniform
  ! decides (exp(-Inf) < 1: vpi_mod.f90:2029-2037) -- and nothing else
  subroutine draw_only(s,w,ng)
    type(sampler_t), intent(inout) :: s
    integer, intent(in) :: w,ng
    integer :: i
    real(8) :: g,u
    do i=1,ng
       call mt_gauss(s%rng(w),g)
    end do
    u = mt_real(s%rng(w))
  end subroutine draw_only

  ! ---- proposal generators (each writes the new bead into the host mirror and queues an item)

  ! free guess of an end bead `ib` from the anchor bead `ia`, variance n*dt
  ! (reference vpi_mod.f90:619-645 and its siblings); side = +1 anchor is "next", -1 "prev"
  subroutine gen_end_guess(s,w,ip,ib,ia,n,side,wgt)
    type(sampler_t), intent(inout) :: s
    integer, intent(in) :: w,ip,ib,ia,n,side
    real(8), intent(in) :: wgt
    real(8) :: xold(s%dim),xnew(s%dim),g,sigma,xm
    integer :: k
    do k=1,s%dim
       xold(k) = s%Path(k,ip,ib,w)
       call mt_gauss(s%rng(w),g)
       if (side>0) then
          xm = image_next(s,k,s%Path(k,ip,ia,w),xold(k))
       else
          xm = image_prev(s,k,s%Path(k,ip,ia,w),xold(k))
       end if
       sigma   = sqrt(dble(n)*s%dt)
       xnew(k) = xm+sigma*g
       call wrap_coord(s,k,xnew(k))
       s%Path(k,ip,ib,w) = xnew(k)
    end do
    call add_item(s,w,ip,ib,xnew,xold,wgt)
  end subroutine gen_end_guess

  ! Levy staging of the interior beads ii+1..ii+L-1 between fixed ends ii and ii+L
  ! (reference vpi_mod.f90:509-549 and its siblings).  Sequential: bead j is drawn around
  ! the NEW bead j-1.
  subroutine gen_staging(s,w,ip,ii,L)
    type(sampler_t), intent(inout) :: s
    integer, intent(in) :: w,ip,ii,L
    real(8) :: xold(s%dim),xnew(s%dim),g,sigma,xp,xn,xm
    integer :: j,k
    do j=1,L-1
       do k=1,s%dim
          xold(k) = s%Path(k,ip,ii+j,w)
          call mt_gauss(s%rng(w),g)
          xp = image_prev(s,k,s%Path(k,ip,ii+j-1,w),xold(k))
          xn = image_next(s,k,s%Path(k,ip,ii+L,w),xold(k))
          ! (real(L-j)/real(L-j+1)) is a single-precision quotient in the reference (Q7)
          sigma   = sqrt(dble(real(L-j)/real(L-j+1))*s%dt)
          xm      = (xn+xp*(L-j))/dble(real(L-j+1))
          xnew(k) = xm+sigma*g
          call wrap_coord(s,k,xnew(k))
          s%Path(k,ip,ii+j,w) = xnew(k)
       end do
       call add_item(s,w,ip,ii+j,xnew,xold,1.d0)
    end do
  end subroutine gen_staging

  ! level `ilev` (1..Nlev) of a bisection of the segment ii..ii+2**Nlev
  ! (reference vpi_mod.f90:905-956)
  subroutine gen_bisection_level(s,w,ip,ii,Nlev,ilev)
    type(sampler_t), intent(inout) :: s
    integer, intent(in) :: w,ip,ii,Nlev,ilev
    real(8) :: xold(s%dim),xnew(s%dim),g,sigma,dt_bis,xp,xn,xm
    integer :: j,k,delta_ib,iprev,inext,icurr
    delta_ib = 2**(Nlev-ilev+1)
    dt_bis   = 0.5d0*dble(real(delta_ib))*s%dt
    sigma    = sqrt(0.5d0*dt_bis)
    do j=1,2**(ilev-1)
       iprev = ii+(j-1)*delta_ib
       inext = ii+j*delta_ib
       icurr = (iprev+inext)/2
       do k=1,s%dim
          xold(k) = s%Path(k,ip,icurr,w)
          call mt_gauss(s%rng(w),g)
          xp = image_prev(s,k,s%Path(k,ip,iprev,w),xold(k))
          xn = image_next(s,k,s%Path(k,ip,inext,w),xold(k))
          xm = 0.5d0*(xp+xn)
          xnew(k) = xm+sigma*g
          call wrap_coord(s,k,xnew(k))
          s%Path(k,ip,icurr,w) = xnew(k)
       end do
       call add_item(s,w,ip,icurr,xnew,xold,1.d0)
    end do
  end subroutine gen_bisection_level

  ! accept/reject of a single-test move for every walker taking part, then device commits
  subroutine settle_simple(s,ip_of,active,accepted,sgn_dk)
    type(sampler_t), intent(inout) :: s
    integer, intent(in)    :: ip_of(s%W)
    logical, intent(in)    :: active(s%W)
    integer, intent(inout) :: accepted(s%W)
    integer, intent(in)    :: sgn_dk            ! 0: exp(-S); -1: exp(-S-DK); +1: exp(-S+DK)
    integer :: w
    real(8) :: t,a
    do w=1,s%W
       s%flag(w) = .false.
       if (.not. active(w)) cycle
       t = sum_items(s,w,s%S0(w))
       if (sgn_dk<0) then
          a = -t-s%DK(w)
       else if (sgn_dk>0) then
          a = -t+s%DK(w)
       else
          a = -t
       end if
       if (metropolis(s,w,a)) then
          accepted(w) = accepted(w)+1
          s%flag(w) = .true.
          call queue_commit_range(s,w,ip_of(w),s%seg_i(w),s%seg_e(w))
       else
          call restore_chain(s,w,ip_of(w))
       end if
    end do
  end subroutine settle_simple

  !=====================================================================
  ! diagonal-sector movers
  !=====================================================================

  ! TranslateChain (reference vpi_mod.f90:313-379): rigid shift of a whole chain
  subroutine mv_translate(s,delta,ip_of,active,accepted)
    type(sampler_t), intent(inout) :: s
    real(8), intent(in)    :: delta
    integer, intent(in)    :: ip_of(s%W)
    logical, intent(in)    :: active(s%W)
    integer, intent(inout) :: accepted(s%W)
    integer :: w,ib,k,ip
    real(8) :: dx(s%dim),xold(s%dim),xnew(s%dim)
    call begin_stage(s)
    where (active) s%want = 2*s%Nb+1
    call plan_items(s)
    !$omp parallel do schedule(static) num_threads(s%nthr) if(s%nthr>1) default(shared) private(w,ip,k,ib,dx,xold,xnew)
    do w=1,s%W
       if (.not. active(w)) cycle
       ip = ip_of(w)
       do k=1,s%dim
          dx(k) = delta*(2.d0*mt_real(s%rng(w))-1.d0)
       end do
       call save_chain(s,w,ip,0,2*s%Nb)
       do ib=0,2*s%Nb
          do k=1,s%dim
             xold(k) = s%Path(k,ip,ib,w)
             xnew(k) = xold(k)+dx(k)
             call wrap_coord(s,k,xnew(k))
             s%Path(k,ip,ib,w) = xnew(k)
          end do
          call add_item(s,w,ip,ib,xnew,xold,1.d0)
       end do
       s%S0(w) = 0.d0
    end do
    !$omp end parallel do
    call evaluate(s)
    call settle_simple(s,ip_of,active,accepted,0)
  end subroutine mv_translate

  ! Bisection (reference vpi_mod.f90:864-998): one Metropolis test per level
  subroutine mv_bisection(s,level,ip_of,active,accepted)
    type(sampler_t), intent(inout) :: s
    integer, intent(in)    :: level
    integer, intent(in)    :: ip_of(s%W)
    logical, intent(in)    :: active(s%W)
    integer, intent(inout) :: accepted(s%W)
    integer :: w,ii
    !$omp parallel do schedule(static) num_threads(s%nthr) if(s%nthr>1) default(shared) private(w,ii)
    do w=1,s%W
       s%alive(w) = active(w)
       if (.not. active(w)) cycle
       ii = int((2*s%Nb-2**level+1)*mt_real(s%rng(w)))
       call save_chain(s,w,ip_of(w),ii,ii+2**level)
       s%aux_i(w) = level
    end do
    !$omp end parallel do
    call run_levels(s,ip_of,active,accepted)
  end subroutine mv_bisection

  ! levels 1..Nlev(w) for the walkers still alive; finishes the move
  subroutine run_levels(s,ip_of,active,accepted)
    type(sampler_t), intent(inout) :: s
    integer, intent(in)    :: ip_of(s%W)
    logical, intent(in)    :: active(s%W)
    integer, intent(inout) :: accepted(s%W)
    integer :: w,ilev,maxlev
    real(8) :: t
    maxlev = 0
    do w=1,s%W
       if (s%alive(w)) maxlev = max(maxlev,s%aux_i(w))
    end do
    do ilev=1,maxlev
       call begin_stage(s)
       do w=1,s%W
          if (s%alive(w) .and. ilev<=s%aux_i(w)) s%want(w) = 2**(ilev-1)
       end do
       call plan_items(s)
       if (s%n_items==0) exit
       !$omp parallel do schedule(static) num_threads(s%nthr) if(s%nthr>1) default(shared) private(w)
       do w=1,s%W
          if (.not. s%alive(w)) cycle
          if (ilev>s%aux_i(w)) cycle
          call gen_bisection_level(s,w,ip_of(w),s%seg_i(w),s%aux_i(w),ilev)
       end do
       !$omp end parallel do
       call evaluate(s)
       !$omp parallel do schedule(static) num_threads(s%nthr) if(s%nthr>1) default(shared) private(w,t)
       do w=1,s%W
          if (.not. s%alive(w)) cycle
          if (ilev>s%aux_i(w)) cycle
          t = sum_items(s,w,0.d0)
          if (.not. metropolis(s,w,-t)) s%alive(w) = .false.
       end do
       !$omp end parallel do
    end do
    do w=1,s%W
       if (.not. active(w)) cycle
       if (s%alive(w)) then
          accepted(w) = accepted(w)+1
          call queue_commit_range(s,w,ip_of(w),s%seg_i(w),s%seg_e(w))
       else
          call restore_chain(s,w,ip_of(w))
       end if
    end do
  end subroutine run_levels

  ! MoveHeadBisection / MoveTailBisection (reference vpi_mod.f90:1002-1372): a free guess of
  ! the end bead with its own Metropolis test (Q12), then bisection levels
  subroutine mv_end_bisection(s,which,level,ip_of,active,accepted)
    type(sampler_t), intent(inout) :: s
    integer, intent(in)    :: which,level
    integer, intent(in)    :: ip_of(s%W)
    logical, intent(in)    :: active(s%W)
    integer, intent(inout) :: accepted(s%W)
    integer :: w,nl,ii,ie
    real(8) :: t
    call begin_stage(s)
    where (active) s%want = 1
    call plan_items(s)
    !$omp parallel do schedule(static) num_threads(s%nthr) if(s%nthr>1) default(shared) private(w,nl,ii,ie)
    do w=1,s%W
       s%alive(w) = active(w)
       if (.not. active(w)) cycle
       nl = int((level-1)*mt_real(s%rng(w)))+2
       s%aux_i(w) = nl
       if (which==HEAD) then
          ii = 0
       else
          ii = 2*s%Nb-2**nl
       end if
       ie = ii+2**nl
       call save_chain(s,w,ip_of(w),ii,ie)
       if (which==HEAD) then
          call gen_end_guess(s,w,ip_of(w),ii,ie,2**nl,+1,1.d0)
       else
          call gen_end_guess(s,w,ip_of(w),ie,ii,2**nl,-1,1.d0)
       end if
    end do
    !$omp end parallel do
    call evaluate(s)
    !$omp parallel do schedule(static) num_threads(s%nthr) if(s%nthr>1) default(shared) private(w,t)
    do w=1,s%W
       if (.not. active(w)) cycle
       t = sum_items(s,w,0.d0)
       if (.not. metropolis(s,w,-t)) s%alive(w) = .false.
    end do
    !$omp end parallel do
    call run_levels(s,ip_of,active,accepted)
  end subroutine mv_end_bisection

  ! Staging (reference vpi_mod.f90:480-578)
  subroutine mv_staging(s,Lstag,ip_of,active,accepted)
    type(sampler_t), intent(inout) :: s
    integer, intent(in)    :: Lstag
    integer, intent(in)    :: ip_of(s%W)
    logical, intent(in)    :: active(s%W)
    integer, intent(inout) :: accepted(s%W)
    integer :: w,ii
    call begin_stage(s)
    where (active) s%want = Lstag-1
    call plan_items(s)
    do w=1,s%W
       if (.not. active(w)) cycle
       ii = int((2*s%Nb-Lstag+1)*mt_real(s%rng(w)))
       call save_chain(s,w,ip_of(w),ii,ii+Lstag)
       call gen_staging(s,w,ip_of(w),ii,Lstag)
       s%S0(w) = 0.d0
    end do
    call evaluate(s)
    call settle_simple(s,ip_of,active,accepted,0)
  end subroutine mv_staging

  ! MoveHead / MoveTail (reference vpi_mod.f90:582-860): free end guess + staging, one test
  subroutine mv_end_staging(s,which,Lmax,ip_of,active,accepted)
    type(sampler_t), intent(inout) :: s
    integer, intent(in)    :: which,Lmax
    integer, intent(in)    :: ip_of(s%W)
    logical, intent(in)    :: active(s%W)
    integer, intent(inout) :: accepted(s%W)
    integer :: w,Ls,ii,ie
    call begin_stage(s)
    do w=1,s%W
       if (.not. active(w)) cycle
       s%aux_i(w) = int((Lmax-1)*mt_real(s%rng(w)))+2
       s%want(w)  = s%aux_i(w)
    end do
    call plan_items(s)
    do w=1,s%W
       if (.not. active(w)) cycle
       Ls = s%aux_i(w)
       if (which==HEAD) then
          ii = 0
       else
          ii = 2*s%Nb-Ls
       end if
       ie = ii+Ls
       call save_chain(s,w,ip_of(w),ii,ie)
       if (which==HEAD) then
          call gen_end_guess(s,w,ip_of(w),ii,ie,Ls,+1,1.d0)
       else
          call gen_end_guess(s,w,ip_of(w),ie,ii,Ls,-1,1.d0)
       end if
       call gen_staging(s,w,ip_of(w),ii,Ls)
       s%S0(w) = 0.d0
    end do
    call evaluate(s)
    call settle_simple(s,ip_of,active,accepted,0)
  end subroutine mv_end_staging

  !=====================================================================
  ! off-diagonal (worm) sector: the open chain iworm has two beads Nb, xend(:,1) closing
  ! the half 0..Nb and xend(:,2) opening the half Nb..2Nb
  !=====================================================================

  ! `Path(:,ip,Nb) = xend(:,half)` at the top of every half-chain mover; the device copy of
  ! that bead is what OTHER particles see at slice Nb, so it is committed as well
  subroutine select_half(s,w,ip,half)
    type(sampler_t), intent(inout) :: s
    integer, intent(in) :: w,ip,half
    if (any(s%Path(:,ip,s%Nb,w)/=s%xend(:,half,w))) then
       s%Path(:,ip,s%Nb,w) = s%xend(:,half,w)
       call queue_commit(s,w,ip,s%Nb,s%Path(:,ip,s%Nb,w))
    end if
  end subroutine select_half

  subroutine settle_half(s,half,active,accepted)
    type(sampler_t), intent(inout) :: s
    integer, intent(in)    :: half
    logical, intent(in)    :: active(s%W)
    integer, intent(inout) :: accepted(s%W)
    integer :: w
    call settle_simple(s,s%iworm,active,accepted,0)
    do w=1,s%W
       if (active(w) .and. s%flag(w)) s%xend(:,half,w) = s%Path(:,s%iworm(w),s%Nb,w)
    end do
  end subroutine settle_half

  ! TranslateHalfChain (reference vpi_mod.f90:383-476)
  subroutine mv_translate_half(s,half,delta,active,accepted)
    type(sampler_t), intent(inout) :: s
    integer, intent(in)    :: half
    real(8), intent(in)    :: delta
    logical, intent(in)    :: active(s%W)
    integer, intent(inout) :: accepted(s%W)
    integer :: w,ib,k,ip,ibi,ibf
    real(8) :: dx(s%dim),xold(s%dim),xnew(s%dim)
    call begin_stage(s)
    where (active) s%want = s%Nb+1
    call plan_items(s)
    do w=1,s%W
       if (.not. active(w)) cycle
       ip = s%iworm(w)
       call select_half(s,w,ip,half)
       do k=1,s%dim
          dx(k) = delta*(2.d0*mt_real(s%rng(w))-1.d0)
       end do
       if (half==1) then
          ibi = 0;    ibf = s%Nb
       else
          ibi = s%Nb; ibf = 2*s%Nb
       end if
       call save_chain(s,w,ip,ibi,ibf)
       do ib=ibi,ibf
          do k=1,s%dim
             xold(k) = s%Path(k,ip,ib,w)
             xnew(k) = xold(k)+dx(k)
             call wrap_coord(s,k,xnew(k))
             s%Path(k,ip,ib,w) = xnew(k)
          end do
          call add_item(s,w,ip,ib,xnew,xold,1.d0)
       end do
       s%S0(w) = 0.d0
    end do
    call evaluate(s)
    call settle_half(s,half,active,accepted)
  end subroutine mv_translate_half

  ! StagingHalfChain (reference vpi_mod.f90:1376-1491)
  subroutine mv_staging_half(s,half,Lstag,active,accepted)
    type(sampler_t), intent(inout) :: s
    integer, intent(in)    :: half,Lstag
    logical, intent(in)    :: active(s%W)
    integer, intent(inout) :: accepted(s%W)
    integer :: w,ii,ip
    logical :: act(s%W)
    call begin_stage(s)
    ! a segment longer than the half chain (Lstag > Nb; refused by the front end in the worm sector, where the
    ! reference indexes outside Path: see seg_outside) is drawn and rejected, never built
    act = active .and. Lstag<=s%Nb
    where (act) s%want = Lstag-1
    call plan_items(s)
    do w=1,s%W
       if (.not. active(w)) cycle
       ip = s%iworm(w)
       call select_half(s,w,ip,half)
       ii = int((s%Nb-Lstag+1)*mt_real(s%rng(w)))
       if (.not. act(w)) then
          call draw_only(s,w,(Lstag-1)*s%dim)
          cycle
       end if
       if (half==2) ii = ii+s%Nb
       call save_chain(s,w,ip,ii,ii+Lstag)
       call gen_staging(s,w,ip,ii,Lstag)
       s%S0(w) = 0.d0
    end do
    call evaluate(s)
    call settle_half(s,half,act,accepted)
  end subroutine mv_staging_half

  ! MoveHeadHalfChain / MoveTailHalfChain (reference vpi_mod.f90:1495-1817); the Delta S of
  ! the cut bead Nb counts half (Q13)
  subroutine mv_end_staging_half(s,which,half,Lmax,active,accepted)
    type(sampler_t), intent(inout) :: s
    integer, intent(in)    :: which,half,Lmax
    logical, intent(in)    :: active(s%W)
    integer, intent(inout) :: accepted(s%W)
    integer :: w,Ls,ii,ie,ip
    real(8) :: wgt
    logical :: act(s%W)
    call begin_stage(s)
    act = active
    do w=1,s%W
       if (.not. active(w)) cycle
       s%aux_i(w) = int((Lmax-1)*mt_real(s%rng(w)))+2
       if (s%aux_i(w)>s%Nb) then
          act(w) = .false.                 ! longer than the half chain: drawn and rejected (see seg_outside)
       else
          s%want(w) = s%aux_i(w)
       end if
    end do
    call plan_items(s)
    do w=1,s%W
       if (.not. active(w)) cycle
       ip = s%iworm(w)
       Ls = s%aux_i(w)
       call select_half(s,w,ip,half)
       if (.not. act(w)) then
          call draw_only(s,w,Ls*s%dim)
          cycle
       end if
       if (which==HEAD) then
          ii = 0
          if (half==2) ii = s%Nb
          wgt = 1.d0
          if (half==2) wgt = 0.5d0
       else
          ii = s%Nb-Ls
          if (half==2) ii = 2*s%Nb-Ls
          wgt = 0.5d0
          if (half==2) wgt = 1.d0
       end if
       ie = ii+Ls
       call save_chain(s,w,ip,ii,ie)
       if (which==HEAD) then
          call gen_end_guess(s,w,ip,ii,ie,Ls,+1,wgt)
       else
          call gen_end_guess(s,w,ip,ie,ii,Ls,-1,wgt)
       end if
       call gen_staging(s,w,ip,ii,Ls)
       s%S0(w) = 0.d0
    end do
    call evaluate(s)
    call settle_half(s,half,act,accepted)
  end subroutine mv_end_staging_half

  ! kinetic weight of the broken link (reference vpi_mod.f90:1872-1873)
  function delta_k(s,r2,Ls) result(dk)
    type(sampler_t), intent(in) :: s
    real(8), intent(in) :: r2
    integer, intent(in) :: Ls
    real(8) :: dk
    dk = -0.5d0*r2/(dble(real(Ls))*s%dt)-0.5d0*dble(real(s%dim))*log(2.d0*s%pi*dble(real(Ls))*s%dt)
  end function delta_k

  ! OpenChain (reference vpi_mod.f90:1821-2076); ip_of(w) is the freshly drawn iworm
  subroutine mv_open(s,Lmax,ip_of,active,accepted)
    type(sampler_t), intent(inout) :: s
    integer, intent(in)    :: Lmax
    integer, intent(in)    :: ip_of(s%W)
    logical, intent(in)    :: active(s%W)
    integer, intent(inout) :: accepted(s%W)
    integer :: w,Ls,half,ii,ie,ip
    logical :: act(s%W)
    call begin_stage(s)
    act = active
    do w=1,s%W
       if (.not. active(w)) cycle
       s%aux_i(w) = 2*int(((Lmax-2)/2)*mt_real(s%rng(w)))+2
       s%aux_k(w) = int(mt_real(s%rng(w))*2)+1
       ! Lstag > Nb: the segment can reach below bead 0 / above bead 2Nb.  Drawn, never built, never accepted
       ! (see seg_outside; the front end refuses Lstag > Nb when CWorm > 0, where the reference could accept it)
       if (s%aux_i(w)>s%Nb) then
          act(w) = .false.
       else
          s%want(w) = s%aux_i(w)
       end if
    end do
    call plan_items(s)
    do w=1,s%W
       if (.not. active(w)) cycle
       ip   = ip_of(w)
       Ls   = s%aux_i(w)
       half = s%aux_k(w)
       if (.not. act(w)) then
          call draw_only(s,w,Ls*s%dim)
          s%flag(w) = .false.
          cycle
       end if
       s%S0(w) = -log(s%CWorm*s%density)
       if (half==1) then
          ii = s%Nb-Ls; ie = s%Nb
       else
          ii = s%Nb;    ie = s%Nb+Ls
       end if
       s%DK(w) = delta_k(s,link_r2(s,s%Path(:,ip,ii,w),s%Path(:,ip,ie,w)),Ls)
       call save_chain(s,w,ip,ii,ie)
       if (half==1) then
          call gen_end_guess(s,w,ip,ie,ii,Ls,-1,0.5d0)
       else
          call gen_end_guess(s,w,ip,ii,ie,Ls,+1,0.5d0)
       end if
       call gen_staging(s,w,ip,ii,Ls)
    end do
    call evaluate(s)
    call settle_simple(s,ip_of,act,accepted,-1)
    do w=1,s%W
       if (.not. active(w)) cycle
       ip = ip_of(w)
       if (s%flag(w)) then
          s%isopen(w) = .true.
          if (s%aux_k(w)==1) then
             s%xend(:,1,w) = s%Path(:,ip,s%Nb,w)
             s%xend(:,2,w) = s%Old(:,s%Nb,w)
          else
             s%xend(:,1,w) = s%Old(:,s%Nb,w)
             s%xend(:,2,w) = s%Path(:,ip,s%Nb,w)
          end if
       else
          s%xend(:,1,w) = s%Path(:,ip,s%Nb,w)
          s%xend(:,2,w) = s%xend(:,1,w)
       end if
    end do
  end subroutine mv_open

  ! CloseChain (reference vpi_mod.f90:2080-2266)
  subroutine mv_close(s,Lmax,active,accepted)
    type(sampler_t), intent(inout) :: s
    integer, intent(in)    :: Lmax
    logical, intent(in)    :: active(s%W)
    integer, intent(inout) :: accepted(s%W)
    integer :: w,Ls,half,ii,ie,ip,ic
    real(8) :: xold(s%dim),xnew(s%dim)
    logical :: act(s%W)
    call begin_stage(s)
    act = active
    do w=1,s%W
       if (.not. active(w)) cycle
       s%aux_i(w) = 2*int(((Lmax-2)/2)*mt_real(s%rng(w)))+2
       s%aux_k(w) = int(mt_real(s%rng(w))*2)+1
       if (s%aux_i(w)>s%Nb) then
          act(w) = .false.                 ! reaches outside 0..2Nb: drawn and rejected (see seg_outside)
       else
          s%want(w) = s%aux_i(w)
       end if
    end do
    call plan_items(s)
    do w=1,s%W
       if (.not. active(w)) cycle
       ip   = s%iworm(w)
       Ls   = s%aux_i(w)
       half = s%aux_k(w)
       if (.not. act(w)) then
          call draw_only(s,w,(Ls-1)*s%dim)
          cycle
       end if
       s%S0(w) = log(s%CWorm*s%density)
       if (half==1) then
          ii = s%Nb-Ls; ie = s%Nb;      ic = ie
       else
          ii = s%Nb;    ie = s%Nb+Ls;   ic = ii
       end if
       call save_chain(s,w,ip,ii,ie)
       ! the cut bead jumps onto the other worm end
       xold = s%Old(:,ic,w)
       xnew = s%xend(:,3-half,w)
       s%Path(:,ip,ic,w) = xnew
       call add_item(s,w,ip,ic,xnew,xold,0.5d0)
       call gen_staging(s,w,ip,ii,Ls)
       s%DK(w) = delta_k(s,link_r2(s,s%Path(:,ip,ii,w),s%Path(:,ip,ie,w)),Ls)
    end do
    call evaluate(s)
    call settle_simple(s,s%iworm,act,accepted,+1)
    do w=1,s%W
       if (.not. active(w)) cycle
       if (s%flag(w)) then
          s%isopen(w) = .false.
          s%xend(:,1,w) = s%Path(:,s%iworm(w),s%Nb,w)
          s%xend(:,2,w) = s%xend(:,1,w)
       end if
    end do
  end subroutine mv_close

  ! Swap (reference vpi_mod.f90:2270-2487).  partner(w) = chosen particle when the swap was
  ! accepted (swapped(w) true)
  subroutine mv_swap(s,Lmax,active,accepted,partner,swapped)
    type(sampler_t), intent(inout) :: s
    integer, intent(in)    :: Lmax
    logical, intent(in)    :: active(s%W)
    integer, intent(inout) :: accepted(s%W)
    integer, intent(out)   :: partner(s%W)
    logical, intent(out)   :: swapped(s%W)
    integer :: w,Ls,ii,ie,ip,ik,iw,ib
    real(8) :: Sw,Sk,uran,acc,t
    real(8), allocatable :: Pp(:)
    logical :: go(s%W)
    allocate(Pp(s%Np))
    swapped = .false.
    partner = 0
    go = .false.
    call begin_stage(s)
    do w=1,s%W
       if (.not. active(w)) cycle
       iw = s%iworm(w)
       Ls = 2*int(((Lmax-2)/2)*mt_real(s%rng(w)))+2
       ii = s%Nb-Ls
       ie = s%Nb
       if (seg_outside(s,ii,ie)) then      ! the partner weights would be read below bead 0: no swap
          uran = mt_real(s%rng(w))
          cycle
       end if
       ! partner selection with Gaussian weights around the worm tail
       Sw = 0.d0
       do ip=1,s%Np
          Pp(ip) = exp(-0.5d0*link_r2(s,s%Path(:,ip,ii,w),s%xend(:,2,w))/(dble(real(Ls))*s%dt))
          Sw     = Sw+Pp(ip)
       end do
       uran = mt_real(s%rng(w))
       ip   = 0
       acc  = 0.d0
       do
          ip  = ip+1
          acc = acc+Pp(ip)/Sw
          if (uran<=acc .or. ip==s%Np) then
             ik = ip
             exit
          end if
       end do
       if (ik==iw) cycle
       Sk = 0.d0
       do ip=1,s%Np
          Sk = Sk+exp(-0.5d0*link_r2(s,s%Path(:,ip,ii,w),s%Path(:,ik,ie,w))/(dble(real(Ls))*s%dt))
       end do
       if (.not. (mt_real(s%rng(w))<=Sw/Sk)) cycle
       go(w) = .true.
       s%aux_k(w) = ik
       s%aux_i(w) = Ls
       s%want(w)  = Ls-1
    end do
    call plan_items(s)
    do w=1,s%W
       if (.not. go(w)) cycle
       iw = s%iworm(w)
       ik = s%aux_k(w)
       Ls = s%aux_i(w)
       ii = s%Nb-Ls
       ie = s%Nb
       call need_beads(s,w,ik,ii,ie)
       call need_beads(s,w,iw,ii,ie)
       s%Old(:,:,w)  = s%Path(:,ik,:,w)
       s%Old2(:,:,w) = s%Path(:,iw,:,w)
       s%seg_i(w) = ii; s%seg_e(w) = ie
       s%Path(:,ik,ie,w) = s%xend(:,2,w)
       call gen_staging(s,w,ik,ii,Ls)
    end do
    call evaluate(s)
    do w=1,s%W
       if (.not. go(w)) cycle
       iw = s%iworm(w)
       ik = s%aux_k(w)
       t  = sum_items(s,w,0.d0)
       if (metropolis(s,w,-t)) then
          accepted(w) = accepted(w)+1
          do ib=s%Nb,2*s%Nb
             s%Path(:,iw,ib,w) = s%Path(:,ik,ib,w)
             s%Path(:,ik,ib,w) = s%Old2(:,ib,w)
          end do
          s%xend(:,2,w)       = s%Old(:,s%Nb,w)
          s%Path(:,iw,s%Nb,w) = s%xend(:,2,w)
          swapped(w) = .true.
          partner(w) = ik
          call queue_commit_range(s,w,ik,s%seg_i(w),2*s%Nb)
          call queue_commit_range(s,w,iw,s%Nb,2*s%Nb)
       else
          s%Path(:,ik,:,w) = s%Old(:,:,w)
          s%Path(:,iw,:,w) = s%Old2(:,:,w)
       end if
    end do
    deallocate(Pp)
  end subroutine mv_swap

end module pigs_sampler
