!-----------------------------------------------------------------------
! pigs_rng -- per-walker random streams of the host sampler.
!
! Each walker owns an MT19937 state and draws exactly the stream the reference
! draws for the same seed (reference random_mod.f90: sgrnd seeds by the 69069
! LCG, grnd() = 32-bit tempered word / (2^32-1) so 1.0 is reachable (quirk Q15),
! rangauss = polar Box-Muller whose second deviate every caller discards).
! Unlike the reference's COMMON block the state is an explicit value, so W
! walkers can be advanced independently (and snapshotted) on one host thread
! or many.
!-----------------------------------------------------------------------
module pigs_rng

  implicit none
  private
  public :: mt_state, mt_seed, mt_real, mt_gauss, mt_save, mt_load

  integer, parameter :: NW = 624, MW = 397

  type mt_state
     integer(4) :: pos = NW+1            ! next word to hand out; NW+1 = never seeded
     integer(4) :: w(0:NW-1) = 0
  end type mt_state

contains

  subroutine mt_seed(s,seed)
    type(mt_state), intent(inout) :: s
    integer(4), intent(in)        :: seed
    integer    :: i
    integer(8) :: t
    s%w(0) = seed
    do i=1,NW-1
       ! 32-bit wrap-around product, done in 64 bits to stay defined
       t      = iand(69069_8*iand(int(s%w(i-1),8),4294967295_8),4294967295_8)
       s%w(i) = to_i32(t)
    end do
    s%pos = NW
  end subroutine mt_seed

  pure function to_i32(t) result(y)
    integer(8), intent(in) :: t
    integer(4) :: y
    if (t>=2147483648_8) then
       y = int(t-4294967296_8,4)
    else
       y = int(t,4)
    end if
  end function to_i32

  ! regenerate the whole block of NW words
  subroutine mt_twist(s)
    type(mt_state), intent(inout) :: s
    integer(4), parameter :: UPPER = ishft(1_4,31), LOWER = not(ishft(1_4,31))
    integer(4), parameter :: MAGIC = int(z'9908B0DF',8)-4294967296_8
    integer    :: i,j
    integer(4) :: y
    if (s%pos==NW+1) call mt_seed(s,4357)
    do i=0,NW-1
       y = ior(iand(s%w(i),UPPER),iand(s%w(mod(i+1,NW)),LOWER))
       j = mod(i+MW,NW)
       s%w(i) = ieor(s%w(j),ishft(y,-1))
       if (btest(y,0)) s%w(i) = ieor(s%w(i),MAGIC)
    end do
    s%pos = 0
  end subroutine mt_twist

  ! uniform in [0,1] (both ends included)
  function mt_real(s) result(u)
    type(mt_state), intent(inout) :: s
    real(8) :: u
    integer(4), parameter :: MASKB = int(z'9D2C5680',8)-4294967296_8
    integer(4), parameter :: MASKC = int(z'EFC60000',8)-4294967296_8
    integer(4) :: y
    if (s%pos>=NW) call mt_twist(s)
    y     = s%w(s%pos)
    s%pos = s%pos+1
    y = ieor(y,ishft(y,-11))
    y = ieor(y,iand(ishft(y,7),MASKB))
    y = ieor(y,iand(ishft(y,15),MASKC))
    y = ieor(y,ishft(y,-18))
    if (y<0) then
       u = (dble(y)+2.0d0**32)/(2.0d0**32-1.0d0)
    else
       u = dble(y)/(2.0d0**32-1.0d0)
    end if
  end function mt_real

  ! first deviate of one polar Box-Muller pair with unit variance; consumes the
  ! same uniforms as the reference's rangauss(1.d0,0.d0,g1,g2)
  subroutine mt_gauss(s,g)
    type(mt_state), intent(inout) :: s
    real(8), intent(out)          :: g
    real(8) :: u1,u2,q
    do
       u1 = 2.d0*mt_real(s)-1.d0
       u2 = 2.d0*mt_real(s)-1.d0
       q  = u1*u1+u2*u2
       if (q<=1.d0) exit
    end do
    q = sqrt((-2.d0*log(q))/q)
    ! mu + sigma*u1*w with mu=0, sigma=1, in the reference's association order
    g = 0.d0+1.d0*u1*q
  end subroutine mt_gauss

  ! generator state on disk: the two unformatted records (index, words) of the reference's
  ! mtsavef / mtgetf (random_mod.f90:125-191).  The reference APPENDS a pair per block and reads the
  ! first pair back (quirk Q10: a resumed run restarts from the OLDEST saved state); we replace the
  ! file, so our own resume continues from the newest state, while a rand_state written by the
  ! reference is read exactly as the reference reads it (first pair).
  subroutine mt_save(s,fname)
    type(mt_state), intent(in)   :: s
    character(len=*), intent(in) :: fname
    integer :: u
    open (newunit=u,file=fname,status='replace',form='unformatted')
    write (u) s%pos
    write (u) s%w
    close (u)
  end subroutine mt_save

  subroutine mt_load(s,fname)
    type(mt_state), intent(inout) :: s
    character(len=*), intent(in)  :: fname
    integer :: u
    open (newunit=u,file=fname,status='old',form='unformatted')
    read (u) s%pos
    read (u) s%w
    close (u)
  end subroutine mt_load

end module pigs_rng
