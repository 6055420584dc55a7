!-----------------------------------------------------------------------
! pigs_host_hooks -- C-callable handles on the host sampler (state in / state out), so that
! its logic can be driven and inspected from outside Fortran: tests compare every mover,
! on identical random-number state, with the reference's own routine.
!-----------------------------------------------------------------------
module pigs_host_hooks

  use iso_c_binding
  use pigs_capi
  use pigs_rng
  use pigs_sampler

  implicit none
  private

  integer, parameter :: MAXS = 16
  type(sampler_t), target, save :: pool(MAXS)
  logical, save :: used(MAXS) = .false.

contains

  function hs_create(dim,Np,Nb,W,trap,dt,density,CWorm,Lbox,ctx) bind(C,name='hs_create') result(h)
    integer(c_int), value :: dim,Np,Nb,W,trap
    real(c_double), value :: dt,density,CWorm
    real(c_double), intent(in) :: Lbox(dim)
    type(c_ptr), value :: ctx
    integer(c_int) :: h
    integer :: i
    h = -1
    do i=1,MAXS
       if (.not. used(i)) then
          used(i) = .true.
          call sampler_init(pool(i),dim,Np,Nb,W,trap/=0,dt,density,CWorm,Lbox,ctx)
          h = i
          return
       end if
    end do
  end function hs_create

  subroutine hs_destroy(h) bind(C,name='hs_destroy')
    integer(c_int), value :: h
    if (h<1 .or. h>MAXS) return
    if (used(h)) call sampler_free(pool(h))
    used(h) = .false.
  end subroutine hs_destroy

  subroutine hs_set_path(h,w,P) bind(C,name='hs_set_path')
    integer(c_int), value :: h,w
    real(c_double), intent(in) :: P(*)
    integer :: n
    n = size(pool(h)%Path(:,:,:,1))
    pool(h)%Path(:,:,:,w+1) = reshape(P(1:n),shape(pool(h)%Path(:,:,:,1)))
  end subroutine hs_set_path

  subroutine hs_get_path(h,w,P) bind(C,name='hs_get_path')
    integer(c_int), value :: h,w
    real(c_double) :: P(*)
    integer :: n
    n = size(pool(h)%Path(:,:,:,1))
    P(1:n) = reshape(pool(h)%Path(:,:,:,w+1),[n])
  end subroutine hs_get_path

  subroutine hs_upload(h) bind(C,name='hs_upload')
    integer(c_int), value :: h
    call sampler_upload(pool(h))
  end subroutine hs_upload

  subroutine hs_flush(h) bind(C,name='hs_flush')
    integer(c_int), value :: h
    call sampler_flush(pool(h))
  end subroutine hs_flush

  subroutine hs_seed(h,w,seed) bind(C,name='hs_seed')
    integer(c_int), value :: h,w,seed
    call mt_seed(pool(h)%rng(w+1),seed)
  end subroutine hs_seed

  subroutine hs_set_rng(h,w,pos,words) bind(C,name='hs_set_rng')
    integer(c_int), value :: h,w,pos
    integer(c_int32_t), intent(in) :: words(0:623)
    pool(h)%rng(w+1)%pos = pos
    pool(h)%rng(w+1)%w   = words
  end subroutine hs_set_rng

  subroutine hs_get_rng(h,w,pos,words) bind(C,name='hs_get_rng')
    integer(c_int), value :: h,w
    integer(c_int) :: pos
    integer(c_int32_t) :: words(0:623)
    pos   = pool(h)%rng(w+1)%pos
    words = pool(h)%rng(w+1)%w
  end subroutine hs_get_rng

  subroutine hs_set_worm(h,w,isopen,iworm,xend) bind(C,name='hs_set_worm')
    integer(c_int), value :: h,w,isopen,iworm
    real(c_double), intent(in) :: xend(*)
    integer :: d
    d = pool(h)%dim
    pool(h)%isopen(w+1) = isopen/=0
    pool(h)%iworm(w+1)  = iworm
    pool(h)%xend(:,:,w+1) = reshape(xend(1:2*d),[d,2])
  end subroutine hs_set_worm

  subroutine hs_get_worm(h,w,isopen,iworm,xend) bind(C,name='hs_get_worm')
    integer(c_int), value :: h,w
    integer(c_int) :: isopen,iworm
    real(c_double) :: xend(*)
    integer :: d
    d = pool(h)%dim
    isopen = merge(1,0,pool(h)%isopen(w+1))
    iworm  = pool(h)%iworm(w+1)
    xend(1:2*d) = reshape(pool(h)%xend(:,:,w+1),[2*d])
  end subroutine hs_get_worm

  ! code: 1 TranslateChain(rpar=delta) 2 Bisection(i1=level) 3 MoveHeadBisection 4 MoveTailBisection
  !       5 Staging(i1=Lstag) 6 MoveHead(i1=Lmax) 7 MoveTail
  !       8 TranslateHalfChain(i2=half,rpar=delta) 9 StagingHalfChain(i1=Lstag,i2=half)
  !       10 MoveHeadHalfChain(i1=Lmax,i2=half) 11 MoveTailHalfChain
  !       12 OpenChain(i1=Lmax) 13 CloseChain(i1=Lmax) 14 Swap(i1=Lmax; partner/swapped out)
  subroutine hs_move(h,code,i1,i2,rpar,ip_of,active,accepted,partner,swapped) bind(C,name='hs_move')
    integer(c_int), value :: h,code,i1,i2
    real(c_double), value :: rpar
    integer(c_int), intent(in) :: ip_of(*),active(*)
    integer(c_int) :: accepted(*),partner(*),swapped(*)
    integer :: W
    integer, allocatable :: ipo(:),acc(:),par(:)
    logical, allocatable :: act(:),swp(:)
    W = pool(h)%W
    allocate(ipo(W),acc(W),par(W),act(W),swp(W))
    ipo = ip_of(1:W); act = active(1:W)/=0; acc = accepted(1:W); par = 0; swp = .false.
    select case (code)
    case (1);  call mv_translate(pool(h),rpar,ipo,act,acc)
    case (2);  call mv_bisection(pool(h),i1,ipo,act,acc)
    case (3);  call mv_end_bisection(pool(h),HEAD,i1,ipo,act,acc)
    case (4);  call mv_end_bisection(pool(h),TAIL,i1,ipo,act,acc)
    case (5);  call mv_staging(pool(h),i1,ipo,act,acc)
    case (6);  call mv_end_staging(pool(h),HEAD,i1,ipo,act,acc)
    case (7);  call mv_end_staging(pool(h),TAIL,i1,ipo,act,acc)
    case (8);  call mv_translate_half(pool(h),i2,rpar,act,acc)
    case (9);  call mv_staging_half(pool(h),i2,i1,act,acc)
    case (10); call mv_end_staging_half(pool(h),HEAD,i2,i1,act,acc)
    case (11); call mv_end_staging_half(pool(h),TAIL,i2,i1,act,acc)
    case (12); call mv_open(pool(h),i1,ipo,act,acc)
    case (13); call mv_close(pool(h),i1,act,acc)
    case (14); call mv_swap(pool(h),i1,act,acc,par,swp)
    end select
    accepted(1:W) = acc
    partner(1:W)  = par
    swapped(1:W)  = merge(1,0,swp)
  end subroutine hs_move

  subroutine hs_uniform_stream(seed,n,u) bind(C,name='hs_uniform_stream')
    integer(c_int), value :: seed,n
    real(c_double) :: u(n)
    type(mt_state) :: st
    integer :: i
    call mt_seed(st,seed)
    do i=1,n
       u(i) = mt_real(st)
    end do
  end subroutine hs_uniform_stream

  subroutine hs_gauss_stream(seed,n,g) bind(C,name='hs_gauss_stream')
    integer(c_int), value :: seed,n
    real(c_double) :: g(n)
    type(mt_state) :: st
    integer :: i
    call mt_seed(st,seed)
    do i=1,n
       call mt_gauss(st,g(i))
    end do
  end subroutine hs_gauss_stream

end module pigs_host_hooks
