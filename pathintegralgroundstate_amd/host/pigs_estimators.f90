!-----------------------------------------------------------------------
! pigs_estimators -- host-side structural estimators, block statistics and the output
! files of the front end (formats of the reference: vpi.f90:517-518, sample_mod.f90:392-932).
! These are O(Nbin) / O(Np^2) once per step (<2 % of the reference's run time, SURVEY §3);
! the O(Np^2 * beads) energy sums are GPU kernels behind pigs_capi.
!-----------------------------------------------------------------------
module pigs_estimators

  implicit none
  private
  public :: est_params, pair_correlation, structure_factor, obdm_accumulate
  public :: normalize_gr, normalize_sk, normalize_nr, variance, perm_state, perm_sampling
  public :: write_radial, write_sk, write_nr

  type est_params
     integer :: dim = 3, Np = 0, Nbin = 100, Nk = 50, Npw = 0
     logical :: trap = .false.
     real(8) :: rcut2 = 0.d0, rbin = 0.d0, pi = 0.d0, CWorm = 0.d0
     real(8) :: Lbox(3) = 1.d0, LboxHalf(3) = 0.5d0, qbin(3) = 0.d0
  end type est_params

  ! permutation-cycle bookkeeping of one walker (reference sample_mod.f90:530-594)
  type perm_state
     logical :: new_cycle = .false., end_cycle = .false.
     integer :: iperm = 1
     integer, allocatable :: members(:), histogram(:)
  end type perm_state

contains

  function image_r2(p,x) result(r2)
    type(est_params), intent(in) :: p
    real(8), intent(inout) :: x(p%dim)
    real(8) :: r2
    integer :: k
    r2 = 0.d0
    do k=1,p%dim
       if (x(k)> p%LboxHalf(k)) x(k) = x(k)-p%Lbox(k)
       if (x(k)<-p%LboxHalf(k)) x(k) = x(k)+p%Lbox(k)
       r2 = r2+x(k)*x(k)
    end do
  end function image_r2

  ! g(r) histogram of one slice: +2 per pair inside the cutoff
  subroutine pair_correlation(p,R,gr)
    type(est_params), intent(in) :: p
    real(8), intent(in)    :: R(p%dim,p%Np)
    real(8), intent(inout) :: gr(p%Nbin)
    real(8) :: x(p%dim),r2
    integer :: i,j,ibin
    do i=1,p%Np-1
       do j=i+1,p%Np
          x  = R(:,i)-R(:,j)
          r2 = image_r2(p,x)
          if (r2<=p%rcut2) then
             ibin     = int(sqrt(r2)/p%rbin)+1
             gr(ibin) = gr(ibin)+2.d0
          end if
       end do
    end do
  end subroutine pair_correlation

  ! S(k) along the box axes, k = iq*2pi/L
  subroutine structure_factor(p,R,Sk)
    type(est_params), intent(in) :: p
    real(8), intent(in)    :: R(p%dim,p%Np)
    real(8), intent(inout) :: Sk(p%dim,p%Nk)
    real(8) :: c,s,qr
    integer :: iq,k,i
    do iq=1,p%Nk
       do k=1,p%dim
          c = 0.d0
          s = 0.d0
          do i=1,p%Np
             qr = real(iq)*p%qbin(k)*R(k,i)
             c  = c+cos(qr)
             s  = s+sin(qr)
          end do
          Sk(k,iq) = Sk(k,iq)+(c*c+s*s)
       end do
    end do
  end subroutine structure_factor

  ! one-body density matrix histogram of the worm's end-to-end vector, with 2m partial waves
  subroutine obdm_accumulate(p,xend,nrho)
    type(est_params), intent(in) :: p
    real(8), intent(in)    :: xend(p%dim,2)
    real(8), intent(inout) :: nrho(0:p%Npw,p%Nbin)
    real(8) :: x(p%dim),r2,r
    complex(8) :: e1,e2,em
    integer :: ibin,m
    x  = xend(:,1)-xend(:,2)
    r2 = image_r2(p,x)
    if (r2<=p%rcut2) then
       r    = sqrt(r2)
       ibin = int(r/p%rbin)+1
       e1   = cmplx(x(1)/r,x(2)/r,8)
       e2   = e1*e1
       em   = 1.d0
       do m=0,p%Npw
          nrho(m,ibin) = nrho(m,ibin)+real(em)
          em = em*e2
       end do
    end if
  end subroutine obdm_accumulate

  ! volume of the unit d-ball
  function unit_ball(p) result(kn)
    type(est_params), intent(in) :: p
    real(8) :: kn
    kn = p%pi**(0.5d0*p%dim)/gamma(0.5d0*p%dim+1.d0)
  end function unit_ball

  subroutine normalize_gr(p,density,ngr,gr)
    type(est_params), intent(in) :: p
    real(8), intent(in)    :: density
    integer, intent(in)    :: ngr
    real(8), intent(inout) :: gr(p%Nbin)
    real(8) :: kn,norm,r,nid
    integer :: ibin
    kn   = unit_ball(p)
    norm = real(p%Np)*real(ngr)
    do ibin=1,p%Nbin
       r   = (real(ibin)-0.5d0)*p%rbin
       nid = density*kn*((r+0.5d0*p%rbin)**p%dim-(r-0.5d0*p%rbin)**p%dim)
       gr(ibin) = gr(ibin)/(nid*norm)
    end do
  end subroutine normalize_gr

  subroutine normalize_sk(p,ngr,Sk)
    type(est_params), intent(in) :: p
    integer, intent(in)    :: ngr
    real(8), intent(inout) :: Sk(p%dim,p%Nk)
    real(8) :: norm
    norm = real(p%Np)*real(ngr)
    Sk = Sk/norm
  end subroutine normalize_sk

  subroutine normalize_nr(p,density,zconf,Nobdm,nrho)
    type(est_params), intent(in) :: p
    real(8), intent(in)    :: density,zconf
    integer, intent(in)    :: Nobdm
    real(8), intent(inout) :: nrho(0:p%Npw,p%Nbin)
    real(8) :: kn,r,nid
    integer :: ibin
    kn = unit_ball(p)
    do ibin=1,p%Nbin
       r   = (real(ibin)-0.5d0)*p%rbin
       nid = density*kn*((r+0.5d0*p%rbin)**p%dim-(r-0.5d0*p%rbin)**p%dim)
       nrho(:,ibin) = nrho(:,ibin)/(p%CWorm*nid*zconf*real(Nobdm))
    end do
  end subroutine normalize_nr

  ! the reference's "variance": standard error sqrt((<x^2>-<x>^2)/n)
  function variance(n,av,av2) result(v)
    integer, intent(in) :: n
    real(8), intent(in) :: av,av2
    real(8) :: v
    v = sqrt((av2-av*av)/real(n))
  end function variance

  subroutine perm_sampling(ps,isopen,iw,ik,swap_accepted)
    type(perm_state), intent(inout) :: ps
    logical, intent(in) :: isopen
    integer, intent(in) :: iw
    integer, intent(in), optional :: ik
    logical, intent(in), optional :: swap_accepted
    logical :: already
    if (ps%new_cycle) then
       ps%members    = 0
       ps%members(1) = iw
       ps%iperm      = 1
       ps%new_cycle  = .false.
    end if
    if (present(swap_accepted)) then
       if (swap_accepted) then
          already = any(ps%members==ik)
          if (.not. ps%end_cycle) then
             if (.not. already) then
                ps%iperm = ps%iperm+1
                ps%members(ps%iperm) = ik
             end if
          end if
       end if
    end if
    if (ps%end_cycle) then
       ps%histogram(ps%iperm) = ps%histogram(ps%iperm)+1
       if (isopen) then
          ps%members    = 0
          ps%members(1) = iw
          ps%iperm      = 1
       end if
       ps%end_cycle = .false.
    end if
  end subroutine perm_sampling

  ! ---- output files (formats of the reference)
  subroutine write_radial(fname,p,n,av,av2)
    character(len=*), intent(in) :: fname
    type(est_params), intent(in) :: p
    integer, intent(in) :: n
    real(8), intent(inout) :: av(p%Nbin),av2(p%Nbin)
    real(8) :: r
    integer :: j,u
    open (newunit=u,file=fname)
    do j=1,p%Nbin
       r      = (real(j)-0.5d0)*p%rbin
       av(j)  = av(j)/real(n)
       av2(j) = av2(j)/real(n)
       write (u,'(20g20.10e3)') r,av(j),variance(n,av(j),av2(j))
    end do
    close (u)
  end subroutine write_radial

  subroutine write_sk(fname,p,n,av,av2)
    character(len=*), intent(in) :: fname
    type(est_params), intent(in) :: p
    integer, intent(in) :: n
    real(8), intent(inout) :: av(p%dim,p%Nk),av2(p%dim,p%Nk)
    integer :: j,k,u
    open (newunit=u,file=fname)
    do j=1,p%Nk
       av(:,j)  = av(:,j)/real(n)
       av2(:,j) = av2(:,j)/real(n)
       write (u,'(20g20.10e3)') (j*p%qbin(k),av(k,j),variance(n,av(k,j),av2(k,j)),k=1,p%dim)
    end do
    close (u)
  end subroutine write_sk

  subroutine write_nr(fname,p,n,av,av2)
    character(len=*), intent(in) :: fname
    type(est_params), intent(in) :: p
    integer, intent(in) :: n
    real(8), intent(inout) :: av(0:p%Npw,p%Nbin),av2(0:p%Npw,p%Nbin)
    real(8) :: r
    integer :: j,m,u
    open (newunit=u,file=fname)
    do j=1,p%Nbin
       r = (real(j)-0.5d0)*p%rbin
       av(:,j)  = av(:,j)/real(n)
       av2(:,j) = av2(:,j)/real(n)
       write (u,'(20g20.10e3)') r,(av(m,j),variance(n,av(m,j),av2(m,j)),m=0,p%Npw)
    end do
    close (u)
  end subroutine write_nr

end module pigs_estimators
