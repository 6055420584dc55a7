!-----------------------------------------------------------------------
! pigs_vpi -- the vpi.in-driven front end on MI355X.
!
!   pigs_vpi < vpi.in
!
! Reads the reference's six namelists from standard input (system, samp, obdm, wavefun,
! extpot, jastrow -- same names, same defaults, reference vpi_mod.f90:14-80,
! system_mod.f90:15-34) plus one optional group of its own,
!     &gpu  n_walkers = 1, device = 0, n_gpus = 1, [device_sampler = T|F,] potential = 'aziz2', k1_variant = 0  /
! (n_gpus = G: the walkers are sharded in contiguous blocks over G GPUs -- devices device .. device+G-1, one
! context and one OpenMP host thread per GPU, no exchange while sampling -- and the block estimators of the shards
! meet ONCE per block in one all-reduce (RCCL over xGMI: pigs_comm_init_all / pigs_estimators_allreduce); thread 0
! writes the walker-summed files.  same_device = T puts every shard on `device`: a one-GPU rehearsal.)
! (potential: aziz2 | lj | dipolar -- the reference selects it by editing system_mod.f90)
! (device_sampler = T: the whole MC step of every walker runs on the GPU in one launch, kernel K6 -- every mover of
! the reference; F: the host-driven lock-step sampler, one K1 batch per move stage.  Left out: K6 wherever it serves
! the input, the host-driven sampler otherwise.  The two give the same files and the same worldlines, bit for bit.)
! runs n_walkers independent PIGS chains in lock-step (walker w uses seed+w-1, so walker 1
! IS the reference's chain), every Delta S / energy sum on the GPU through libpigs_hip.so,
! and writes the reference's observable files: e_vpi.out, et_vpi.out ('(5g20.10e3)'),
! gr_vpi.out, sk_vpi.out, nr_vpi.out ('(20g20.10e3)'), fort.99-style permutation histogram
! (perm_vpi.out), plus e_vpi.hex / et_vpi.hex with the same numbers as 64-bit hex for
! parity checks beyond 10 digits.  With n_walkers > 1 the per-walker files carry a .wNNNN
! suffix and the unsuffixed files hold the walker average.
! The step / block structure follows reference vpi.f90:244-588.
!-----------------------------------------------------------------------
program pigs_vpi

  use iso_c_binding
  use pigs_capi
  use pigs_rng
  use pigs_sampler
  use pigs_estimators
  use omp_lib

  implicit none

  ! ---- namelist variables (names fixed by the input format)
  logical           :: crystal,trap,resume,swapping,wf_table,v_table
  character (len=3) :: sampling
  real (kind=8)     :: density,dt,delta_cm,CWorm,Rm
  real (kind=8)     :: a_ho(3)
  integer           :: dim,Np,Nb,seed,CMFreq,Lstag,Nlev,Nstag,Nblock,Nstep,Nbin,Nk
  integer           :: Nobdm,Npw,Nmax,n_walkers,device,ios,k1_variant,n_gpus
  logical           :: device_sampler,checkpointing,same_device
  logical           :: sampler_auto
  integer(c_int)    :: rc_probe
  type(pigs_sweep_params) :: probe_par
  character (len=8) :: potential
  integer           :: pot_kind
  namelist /system/  dim,Np,density,crystal,trap
  namelist /samp/    resume,dt,Nb,seed,delta_cm,CMFreq,sampling,Lstag,Nlev,Nstag,Nblock,Nstep,Nbin,Nk
  namelist /obdm/    swapping,CWorm,Nobdm,Npw
  namelist /wavefun/ Nmax,wf_table,v_table
  namelist /extpot/  a_ho
  namelist /jastrow/ Rm
  namelist /gpu/     n_walkers,device,device_sampler,potential,checkpointing,k1_variant,n_gpus,same_device


  ! shared by the shards (read-only once the parallel region starts)
  type(pigs_params)  :: gp
  type(c_ptr), allocatable :: ctxs(:)
  integer, allocatable :: lo(:),hi(:)
  real(8) :: Lbox(3),rcut,rcut2,rbin,dr,pi
  real(8), allocatable :: VTable(:),LogWF(:)
  integer :: NWtot,G,ish0,k,ucfg0
  real(8), allocatable :: chk_all(:,:,:,:),AE_all(:,:),AT_all(:,:)
  integer, allocatable :: diag_bl_all(:)

  !---------------------------------------------------------------------
  ! defaults (reference vpi_mod.f90:39-60) and input
  crystal = .false.; trap = .false.
  resume = .false.; seed = 1982; Lstag = 2; Nlev = 1
  swapping = .false.; CWorm = 0.d0; Nobdm = 0; Npw = 0
  Nmax = 10000; wf_table = .false.; v_table = .false.
  CMFreq = 1; Nstag = 1; Nblock = 1; Nstep = 1; Nbin = 100; Nk = 50; sampling = 'bis'
  delta_cm = 0.d0; density = 0.d0; a_ho = 1.d0; Rm = 1.d0
  n_walkers = 1; device = 0; device_sampler = .false.; potential = 'aziz2'; checkpointing = .true.; k1_variant = 0
  n_gpus = 1; same_device = .false.

  read (5,nml=system,iostat=ios);  rewind (5)
  read (5,nml=samp,iostat=ios);    rewind (5)
  read (5,nml=obdm,iostat=ios);    rewind (5)
  read (5,nml=wavefun,iostat=ios); rewind (5)
  if (trap) then
     read (5,nml=extpot,iostat=ios); rewind (5)
  end if
  read (5,nml=jastrow,iostat=ios); rewind (5)
  ! device_sampler left out of &gpu = automatic: the device-resident sampler (K6) wherever it serves the input, the
  ! host-driven one otherwise.  Whether it was given is found by reading the group twice with opposite defaults.
  device_sampler = .true.
  read (5,nml=gpu,iostat=ios);     rewind (5)
  sampler_auto = device_sampler
  device_sampler = .false.
  read (5,nml=gpu,iostat=ios);     rewind (5)
  sampler_auto = sampler_auto .neqv. device_sampler

  if (.not. v_table) then
     write (0,*) 'pigs_vpi: v_table = T is required (the reference Force() is a stub: system_mod.f90:186-209)'
     stop 2
  end if
  NWtot = n_walkers
  G = max(1,min(n_gpus,NWtot))
  pi = acos(-1.d0)
  if (Lstag>Nb .and. CWorm>0.d0) then
     ! the worm-sector movers cut segments of up to Lstag beads out of a HALF chain: beyond Nb the reference indexes
     ! Path below bead 0 / above bead 2Nb (vpi_mod.f90:1853-1857, 2112-2116, 2300) and may even accept what it read
     ! there.  With CWorm = 0 the open proposal is never accepted (quirk Q11) and both samplers only draw its
     ! random numbers; with CWorm > 0 there is no defined result to reproduce: refuse the input.
     write (0,*) 'pigs_vpi: CWorm > 0 needs Lstag <= Nb (worm moves act on half chains of Nb links)'
     stop 2
  end if

  ! box, cutoff, grids (reference vpi.f90:82-128)
  Lbox = 1.d0
  if (trap) then
     rcut = 1.d0
     do k=1,dim
        rcut = 3.d0*rcut*a_ho(k)
     end do
     density  = real(Np)/(pi**(0.5d0*dim)*rcut/gamma(0.5d0*dim+1.d0))
     rcut     = rcut**(1.d0/real(dim))
     rcut     = 10.d0*rcut
     delta_cm = delta_cm*minval(a_ho(1:dim))
  else
     if (crystal) then
        ! lattice start: particle number, box and density come from config_ini.in (reference vpi.f90:99-107)
        open (newunit=ucfg0,file='config_ini.in',status='old')
        read (ucfg0,*) Np
        read (ucfg0,*) (Lbox(k),k=1,dim)
        read (ucfg0,*) density
        close (ucfg0)
     else
        do k=1,dim
           Lbox(k) = (real(Np)/density)**(1.d0/real(dim))
        end do
     end if
     rcut     = minval(0.5d0*Lbox(1:dim))
     delta_cm = delta_cm/density**(1.d0/real(dim))
  end if
  rcut2 = rcut*rcut
  rbin  = rcut/real(Nbin)

  ! tables on the host (reference vpi_mod.f90:84-145), then the GPU context
  allocate (VTable(0:Nmax+1),LogWF(0:Nmax+1))
  ! the reference picks its pair potential by editing system_mod.f90; here it is an input
  select case (trim(potential))
  case ('aziz2');   pot_kind = 0
  case ('lj');      pot_kind = 1
  case ('dipolar'); pot_kind = 2
  case default
     write (0,*) 'pigs_vpi: unknown potential ',trim(potential),' (aziz2 | lj | dipolar)'
     stop 2
  end select
  call pigs_check(pigs_build_tables_kind(int(pot_kind,c_int32_t),int(Nmax,c_int32_t),Rm,rcut,VTable,LogWF,dr), &
       & 'pigs_build_tables_kind')
  gp%dim = dim; gp%Np = Np; gp%Nb = Nb; gp%Nmax = Nmax
  gp%trap = merge(1,0,trap); gp%wf_table = merge(1,0,wf_table); gp%v_table = 1; gp%reserved = 0
  gp%dr = dr; gp%rcut2 = rcut2; gp%dt = dt; gp%Rm = Rm
  gp%Lbox = Lbox; gp%a_ho = a_ho
  ! one context per GPU, each with a contiguous block of walkers (sharding.shard_walkers: the first mod(NW,G) shards
  ! get one walker more); walker w (global, 1-based) runs the chain of seed+w-1 whatever the partition
  allocate (ctxs(G),lo(G),hi(G))
  do ish0=1,G
     lo(ish0) = (ish0-1)*(NWtot/G)+min(ish0-1,mod(NWtot,G))+1
     hi(ish0) = lo(ish0)+NWtot/G-1
     if (ish0<=mod(NWtot,G)) hi(ish0) = hi(ish0)+1
     call pigs_check(pigs_ctx_create(gp,VTable,LogWF,int(hi(ish0)-lo(ish0)+1,c_int32_t), &
          & int(merge(device,device+ish0-1,same_device),c_int32_t),ctxs(ish0)),'pigs_ctx_create')
     ! k1_variant = 2 selects the Delta-S kernel that keeps the reference's rounding of every term (default 0: short arithmetic)
     if (k1_variant/=0) call pigs_check(pigs_set_tuning(ctxs(ish0),'k1_variant'//c_null_char,int(k1_variant,c_int32_t)), &
          & 'pigs_set_tuning')
  end do
  if (G>1) call pigs_check(pigs_comm_init_all(ctxs,int(G,c_int32_t)),'pigs_comm_init_all')
  if (sampler_auto) then
     ! both samplers give the same files and the same worldlines, bit for bit (soaked against each other and against the
     ! reference's arithmetic: profiles/r03_k6_vs_host_soak.txt); the device-resident one is ~30 times faster at N = 256
     call fill_sweep_params(probe_par)
     rc_probe = pigs_sampler_init(ctxs(1),probe_par)
     device_sampler = rc_probe==PIGS_OK
  end if
  allocate (chk_all(dim,Np,0:2*Nb,NWtot),AE_all(3,NWtot),AT_all(3,NWtot),diag_bl_all(NWtot))
  diag_bl_all = 0

  print '(a)',       ' =============================================================='
  print '(a)',       '            VPI Monte Carlo on MI355X (pigs_vpi)               '
  print '(a)',       ' =============================================================='
  print '(a,i6)',    '  > Walkers (lock-step) :',NWtot
  print '(a,i6)',    '  > GPUs (walker shards):',G
  print '(a,i6)',    '  > Dimensions          :',dim
  print '(a,i6)',    '  > Number of particles :',Np
  print '(a,i6)',    '  > Number of beads     :',Nb
  print '(a,g13.6)', '  > Time step           :',dt
  print '(a,i6)',    '  > Number of blocks    :',Nblock
  print '(a,i6)',    '  > MC steps per block  :',Nstep
  if (device_sampler) then
     print '(a)',    '  > Sampler             : device-resident (K6: one launch per MC step)'
  else if (sampler_auto) then
     print '(a)',    '  > Sampler             : host-driven (the device-resident sampler does not serve this input or backend)'
  else
     print '(a)',    '  > Sampler             : host-driven (lock-step batches through K1)'
  end if

  !=====================================================================

  !$omp parallel num_threads(G) default(shared)
  call run_shard(omp_get_thread_num()+1)
  !$omp end parallel

  print '(a)', ' =============================================================='
  print '(a)', ' FINAL RESULTS (per walker: <E> <Ec> <Ep> | <Et> <Kt> <Vt>, per particle)'
  do k=1,NWtot
     if (diag_bl_all(k)>0) print '(i6,6g16.8)', k-1,AE_all(:,k)/real(diag_bl_all(k))/Np,AT_all(:,k)/real(diag_bl_all(k))/Np
  end do
  print '(a)', ' =============================================================='
  open (newunit=k,file='worldlines_final.bin',form='unformatted',access='stream')
  write (k) chk_all
  close (k)
  do ish0=1,G
     call pigs_check(pigs_ctx_destroy(ctxs(ish0)),'pigs_ctx_destroy')
  end do

contains

  ! namelists samp + obdm as the library's sweep parameters
  subroutine fill_sweep_params(p)
    type(pigs_sweep_params), intent(out) :: p
    p%Nlev = Nlev; p%Nstag = Nstag; p%CMFreq = CMFreq; p%Lstag = Lstag
    p%delta_cm = delta_cm
    p%CWorm = CWorm; p%density = density; p%rbin = rbin
    p%sampling = merge(1,0,sampling=="sta")
    p%swapping = merge(1,0,swapping); p%Nobdm = Nobdm; p%Nbin = Nbin; p%Npw = Npw
    p%reserved = 0
  end subroutine fill_sweep_params

  !---------------------------------------------------------------------
  ! one shard: the walkers lo(ish)..hi(ish) on context ctxs(ish), the reference's block / step structure
  ! (vpi.f90:244-588) for all of them in lock-step.  Everything below is private to the calling thread.
  subroutine run_shard(ish)
  integer, intent(in) :: ish
  integer(c_int32_t) :: rpos
  integer :: NW,w0
  logical :: trace
  character(len=8) :: envbuf

  type(sampler_t)    :: s
  type(est_params)   :: ep
  type(c_ptr)        :: ctx
  type(perm_state), allocatable :: perm(:)

  integer :: w,k,ip,ib,istep,iblock,istag,iobdm,j,nd,i,nvec,ndall,ngrall,nnrall,ngrav,nnrav
  integer, allocatable :: ipv(:),iupd(:),partner(:),diag_list(:)
  logical, allocatable :: act(:),isopen0(:),swp(:)
  integer(c_int32_t), allocatable :: wl(:)
  real(8), allocatable :: E1(:),E2(:),K1(:),P1(:),Et(:),Kt(:),Pt(:)
  real(8) :: E,Kin,Pot

  ! per-walker counters and accumulators (one column per walker)
  integer, allocatable :: acc_cm(:),acc_bd(:),acc_head(:),acc_tail(:),acc_cm_half(:),acc_bd_half(:)
  integer, allocatable :: acc_head_half(:),acc_tail_half(:),acc_open(:),acc_close(:),acc_swap(:)
  integer, allocatable :: try_open(:),try_close(:),try_swap(:)
  real(8), allocatable :: try_cm(:),try_stag(:),try_cm_half(:),try_stag_half(:)
  integer, allocatable :: idiag(:),idiag_aux(:),idiag_block(:),obdm_bl(:),diag_bl(:),ngr(:)
  real(8), allocatable :: BE(:,:),BE2(:,:),BT(:,:),BT2(:,:),AE(:,:),AE2(:,:),AT(:,:),AT2(:,:)
  real(8), allocatable :: gr(:,:),AvGr(:,:),AvGr2(:,:),Sk(:,:,:),AvSk(:,:,:),AvSk2(:,:,:)
  real(8), allocatable :: nrho(:,:,:),AvNr(:,:,:),AvNr2(:,:,:)
  real(8) :: t0,t1,mE(3),mT(3)
  integer, allocatable :: ue(:),ut(:),uh(:)
  integer :: ueav,utav,ucfg
  logical :: lflag
  integer(8) :: c0,c1,crate
  type(pigs_sweep_params) :: swp_par
  real(8), allocatable, target :: gr_inc(:,:),sk_inc(:,:,:)
  real(8), allocatable :: en9(:,:)
  logical :: est_pending,est_have,pend_struct
  integer :: pend_nd
  integer, allocatable :: pend_list(:)
  integer(c_int64_t), allocatable :: dev_acc(:,:),dev_acc0(:,:)
  integer(c_int32_t), allocatable :: dev_open(:),dev_iworm(:),dev_ev(:,:),dev_reset(:)
  real(8), allocatable :: dev_nrho(:,:,:)
  character(len=32) :: suffix

  real(8), allocatable :: vec(:),AvGrAll(:),AvGr2All(:),AvSkAll(:,:),AvSk2All(:,:),AvNrAll(:,:),AvNr2All(:,:),tmp1(:),tmp2(:,:),tmp3(:,:)
  real(8) :: cnt_all(13)

  call get_environment_variable('PIGS_VPI_TRACE',envbuf)
  trace = envbuf(1:1)=='1'
  NW  = hi(ish)-lo(ish)+1
  w0  = lo(ish)-1                        ! global walker index = w0 + local index
  ctx = ctxs(ish)

  call sampler_init(s,dim,Np,Nb,NW,trap,dt,density,CWorm,Lbox(1:dim),ctx)
  if (ish==1 .and. .not. device_sampler) print '(a,i6)','  > host threads per stage:',s%nthr
  ep%dim = dim; ep%Np = Np; ep%Nbin = Nbin; ep%Nk = Nk; ep%Npw = Npw; ep%trap = trap
  ep%rcut2 = rcut2; ep%rbin = rbin; ep%pi = pi; ep%CWorm = CWorm
  ep%Lbox = Lbox; ep%LboxHalf = 0.5d0*Lbox; ep%qbin = 2.d0*pi/Lbox

  ! initial configuration (reference vpi_mod.f90:149-259): resume from checkpoint files, a lattice
  ! from config_ini.in, or uniform random positions; every bead of a particle starts at the same
  ! point; walker w seeds its stream with seed+w-1
  do w=1,NW
     suffix = ''
     if (NWtot>1) write (suffix,'(a,i4.4)') '.w',w0+w-1
     if (resume) then
        open (newunit=ucfg,file='checkpoint'//trim(suffix)//'.dat',status='old')
        read (ucfg,*) lflag                      ! trap: the reference's init takes it from the file (vpi_mod.f90:166);
        if (lflag .neqv. trap) then              ! here the tables and the box were already built from the namelist value,
           write (0,*) 'pigs_vpi: checkpoint',trim(suffix),'.dat was written with trap = ',lflag, &   ! so a mismatch is an error
                & ' but the namelist says trap = ',trap
           stop 2
        end if
        read (ucfg,*) lflag
        s%isopen(w) = lflag
        read (ucfg,*) s%iworm(w)
        do ip=1,Np
           do ib=0,2*Nb
              read (ucfg,*) (s%Path(k,ip,ib,w),k=1,dim)
           end do
        end do
        read (ucfg,*)
        read (ucfg,*)
        do j=1,2
           read (ucfg,*) (s%xend(k,j,w),k=1,dim)
        end do
        close (ucfg)
        call mt_load(s%rng(w),'rand_state'//trim(suffix))
        cycle
     end if
     call mt_seed(s%rng(w),seed+w0+w-1)
     if (crystal .and. .not. trap) then
        open (newunit=ucfg,file='config_ini.in',status='old')
        read (ucfg,*)
        read (ucfg,*)
        read (ucfg,*)
        do ip=1,Np
           read (ucfg,*) (s%Path(k,ip,0,w),k=1,dim)
        end do
        close (ucfg)
     else
        do ip=1,Np
           do k=1,dim
              if (trap) then
                 s%Path(k,ip,0,w) = 2.d0*a_ho(k)*(mt_real(s%rng(w))-0.5d0)
              else
                 s%Path(k,ip,0,w) = Lbox(k)*(mt_real(s%rng(w))-0.5d0)
              end if
           end do
        end do
     end if
     do ib=1,2*Nb
        s%Path(:,:,ib,w) = s%Path(:,:,0,w)
     end do
     s%xend(:,1,w) = s%Path(:,Np,Nb,w)
     s%xend(:,2,w) = s%xend(:,1,w)
  end do
  call sampler_upload(s)
  if (device_sampler) then
     call fill_sweep_params(swp_par)
     call pigs_check(pigs_sampler_init(ctx,swp_par),'pigs_sampler_init')
     call pigs_check(pigs_sampler_event_ints(ctx,rpos),'pigs_sampler_event_ints')
     allocate (dev_open(NW),dev_iworm(NW),dev_ev(rpos,NW),dev_nrho(0:Npw,Nbin,NW),dev_reset(NW))
     dev_open = merge(1,0,s%isopen); dev_iworm = s%iworm
     call pigs_check(pigs_sampler_set_worm(ctx,dev_open,dev_iworm,s%xend),'pigs_sampler_set_worm')
     do w=1,NW
        call pigs_check(pigs_sampler_set_rng(ctx,int(w-1,c_int32_t),int(s%rng(w)%pos,c_int32_t),s%rng(w)%w), &
             & 'pigs_sampler_set_rng')
     end do
     allocate (gr_inc(Nbin,NW),sk_inc(dim,Nk,NW),dev_acc(16,NW),dev_acc0(16,NW))
     dev_acc0 = 0
  end if

  allocate (perm(NW))
  do w=1,NW
     allocate (perm(w)%members(Np),perm(w)%histogram(Np))
     perm(w)%members = 0; perm(w)%histogram = 0
  end do

  allocate (ipv(NW),iupd(NW),partner(NW),diag_list(NW),act(NW),isopen0(NW),swp(NW),wl(NW))
  allocate (E1(NW),E2(NW),K1(NW),P1(NW),Et(NW),Kt(NW),Pt(NW),en9(9,NW),pend_list(NW))
  est_pending = .false.; est_have = .false.; pend_struct = .false.; pend_nd = 0
  allocate (acc_cm(NW),acc_bd(NW),acc_head(NW),acc_tail(NW),acc_cm_half(NW),acc_bd_half(NW))
  allocate (acc_head_half(NW),acc_tail_half(NW),acc_open(NW),acc_close(NW),acc_swap(NW))
  allocate (try_open(NW),try_close(NW),try_swap(NW),try_cm(NW),try_stag(NW),try_cm_half(NW),try_stag_half(NW))
  allocate (idiag(NW),idiag_aux(NW),idiag_block(NW),obdm_bl(NW),diag_bl(NW),ngr(NW))
  allocate (BE(3,NW),BE2(3,NW),BT(3,NW),BT2(3,NW),AE(3,NW),AE2(3,NW),AT(3,NW),AT2(3,NW))
  allocate (gr(Nbin,NW),AvGr(Nbin,NW),AvGr2(Nbin,NW),Sk(dim,Nk,NW),AvSk(dim,Nk,NW),AvSk2(dim,Nk,NW))
  allocate (nrho(0:Npw,Nbin,NW),AvNr(0:Npw,Nbin,NW),AvNr2(0:Npw,Nbin,NW))
  AE = 0.d0; AE2 = 0.d0; AT = 0.d0; AT2 = 0.d0
  AvGr = 0.d0; AvGr2 = 0.d0; AvSk = 0.d0; AvSk2 = 0.d0; AvNr = 0.d0; AvNr2 = 0.d0; nrho = 0.d0
  idiag = 0; idiag_aux = 0; obdm_bl = 0; diag_bl = 0

  allocate (ue(NW),ut(NW),uh(NW))
  do w=1,NW
     suffix = ''
     if (NWtot>1) write (suffix,'(a,i4.4)') '.w',w0+w-1
     open (newunit=ue(w),file='e_vpi'//trim(suffix)//'.out')
     open (newunit=ut(w),file='et_vpi'//trim(suffix)//'.out')
     open (newunit=uh(w),file='e_vpi'//trim(suffix)//'.hex')
  end do
  if (NWtot>1 .and. ish==1) then
     open (newunit=ueav,file='e_vpi.out')
     open (newunit=utav,file='et_vpi.out')
  end if

  ! the vector that meets the other shards' once per block: number of walkers with a diagonal block, their summed block
  ! energies, the block's counters, the summed normalised g(r), S(k), n(r) and how many walkers contributed to each
  nvec = 7+13+Nbin+dim*Nk+(Npw+1)*Nbin+2
  allocate (vec(nvec),AvGrAll(Nbin),AvGr2All(Nbin),AvSkAll(dim,Nk),AvSk2All(dim,Nk),AvNrAll(0:Npw,Nbin),AvNr2All(0:Npw,Nbin))
  allocate (tmp1(Nbin),tmp2(dim,Nk),tmp3(0:Npw,Nbin))
  AvGrAll = 0.d0; AvGr2All = 0.d0; AvSkAll = 0.d0; AvSk2All = 0.d0; AvNrAll = 0.d0; AvNr2All = 0.d0
  ngrav = 0; nnrav = 0

  do iblock=1,Nblock

     call system_clock(c0,crate)
     try_open = 0; acc_open = 0; try_close = 0; acc_close = 0
     try_cm = 0; acc_cm = 0; try_stag = 0; acc_bd = 0; acc_head = 0; acc_tail = 0
     try_cm_half = 0; acc_cm_half = 0; try_stag_half = 0
     acc_bd_half = 0; acc_head_half = 0; acc_tail_half = 0
     try_swap = 0; acc_swap = 0
     idiag_block = 0; ngr = 0; gr = 0.d0; Sk = 0.d0
     BE = 0.d0; BE2 = 0.d0; BT = 0.d0; BT2 = 0.d0

     ! one trip more than steps: the estimators of a step are collected at the top of the NEXT trip -- with the
     ! device-resident sampler they run on the context's second stream, on a snapshot of the worldlines, while that next step
     ! is sampled (pigs_diagonal_estimators_begin / _end: at 128 walkers per GPU the sampler leaves half of the CUs idle)
     do istep=1,Nstep+1

        if (device_sampler .and. istep<=Nstep) then
           ! the whole step of every walker in one launch (K6); queued, not waited for
           call pigs_check(pigs_sampler_step(ctx,int(istep,c_int32_t)),'pigs_sampler_step')
        end if
        ! ---- the previous step's estimators (reference vpi.f90:443-473)
        if (est_pending) then
           if (pend_struct) then
              call pigs_check(pigs_diagonal_estimators_end(ctx,en9,c_loc(gr_inc),c_loc(sk_inc)),'pigs_diagonal_estimators_end')
           else
              call pigs_check(pigs_diagonal_estimators_end(ctx,en9,c_null_ptr,c_null_ptr),'pigs_diagonal_estimators_end')
           end if
           est_pending = .false.
           est_have = .true.
        end if
        if (est_have) then
           est_have = .false.
           do i=1,pend_nd
              E1(i) = en9(1,i); E2(i) = en9(4,i); Et(i) = en9(7,i); Kt(i) = en9(8,i); Pt(i) = en9(9,i)
           end do
           !$omp parallel do schedule(dynamic,1) private(w,E,Pot,Kin) num_threads(min(pend_nd,16))
           do i=1,pend_nd
              w = pend_list(i)
              idiag(w) = idiag(w)+1; idiag_aux(w) = idiag_aux(w)+1; idiag_block(w) = idiag_block(w)+1
              E   = 0.5d0*(E1(i)+E2(i))
              Pot = Pt(i)
              Kin = E-Pot
              BE(:,w)  = BE(:,w)+[E,Kin,Pot]
              BT(:,w)  = BT(:,w)+[Et(i),Kt(i),Pot]
              BE2(:,w) = BE2(:,w)+[E**2,Kin**2,Pot**2]
              BT2(:,w) = BT2(:,w)+[Et(i)**2,Kt(i)**2,Pot**2]
              ngr(w) = ngr(w)+1
              if (.not. trap) then
                 if (device_sampler) then
                    gr(:,w)   = gr(:,w)+gr_inc(:,i)
                    Sk(:,:,w) = Sk(:,:,w)+sk_inc(:,:,i)
                 else
                    ! (host-driven sampler: the mirror still holds the step's worldline -- the next step's moves come below)
                    call pair_correlation(ep,s%Path(:,:,Nb,w),gr(:,w))
                    call structure_factor(ep,s%Path(:,:,Nb,w),Sk(:,:,w))
                 end if
              end if
           end do
           !$omp end parallel do
        end if
        if (istep>Nstep) exit

        if (device_sampler) then
           if (CWorm>0.d0) then
              ! sector of every walker after the step and what its worm did during it: the permutation-cycle
              ! bookkeeping of the reference (sample_mod.f90:530-594) is replayed from the event log
              call pigs_check(pigs_sampler_events(ctx,dev_ev),'pigs_sampler_events')
              do w=1,NW
                 s%isopen(w) = dev_ev(2,w)/=0
                 if (swapping) then
                    do i=1,dev_ev(1,w)
                       select case (dev_ev(1+2*i,w))
                       case (1)
                          s%iworm(w) = dev_ev(2+2*i,w)
                          perm(w)%new_cycle = .true.
                          call perm_sampling(perm(w),.true.,s%iworm(w))
                       case (2)
                          perm(w)%end_cycle = .true.
                          call perm_sampling(perm(w),.false.,s%iworm(w))
                       case (3)
                          call perm_sampling(perm(w),.true.,s%iworm(w),dev_ev(2+2*i,w),.true.)
                       end select
                    end do
                 end if
              end do
           end if
        else
        ! ---- open / close attempt (reference vpi.f90:302-323)
        isopen0 = s%isopen
        do w=1,NW
           iupd(w) = int(mt_real(s%rng(w))*2)
        end do
        act = isopen0 .and. iupd==0
        if (any(act)) then
           call mv_close(s,Lstag,act,acc_close)
           do w=1,NW
              if (.not. act(w)) cycle
              perm(w)%end_cycle = .not. s%isopen(w)
              try_close(w) = try_close(w)+1
              if (swapping) call perm_sampling(perm(w),s%isopen(w),s%iworm(w))
           end do
        end if
        act = (.not. isopen0) .and. iupd==1
        if (any(act)) then
           do w=1,NW
              if (act(w)) s%iworm(w) = min(int(mt_real(s%rng(w))*Np)+1,Np)
           end do
           call mv_open(s,Lstag,s%iworm,act,acc_open)
           do w=1,NW
              if (.not. act(w)) cycle
              perm(w)%new_cycle = s%isopen(w)
              try_open(w) = try_open(w)+1
              if (swapping) call perm_sampling(perm(w),s%isopen(w),s%iworm(w))
           end do
        end if

        ! ---- centre-of-mass and bead moves of every (non-worm) particle
        if (mod(istep,CMFreq)==0) then
           do ip=1,Np
              ipv = ip
              act = .not. (s%isopen .and. s%iworm==ip)
              where (act) try_cm = try_cm+1
              call mv_translate(s,delta_cm,ipv,act,acc_cm)
           end do
        end if
        do istag=1,Nstag
           do ip=1,Np
              ipv = ip
              act = .not. (s%isopen .and. s%iworm==ip)
              where (act) try_stag = try_stag+1
              if (sampling=="sta") then
                 call mv_end_staging(s,HEAD,Lstag,ipv,act,acc_head)
                 call mv_end_staging(s,TAIL,Lstag,ipv,act,acc_tail)
                 call mv_staging(s,Lstag,ipv,act,acc_bd)
              else
                 call mv_end_bisection(s,HEAD,Nlev,ipv,act,acc_head)
                 call mv_end_bisection(s,TAIL,Nlev,ipv,act,acc_tail)
                 call mv_bisection(s,Nlev,ipv,act,acc_bd)
              end if
           end do
        end do

        ! ---- worm moves of the walkers in the off-diagonal sector (reference vpi.f90:370-404)
        act = s%isopen
        if (any(act)) then
           do iobdm=1,Nobdm
              do j=1,2
                 where (act) try_cm_half = try_cm_half+1
                 call mv_translate_half(s,j,delta_cm,act,acc_cm_half)
              end do
              do j=1,2
                 where (act) try_stag_half = try_stag_half+1
                 call mv_end_staging_half(s,HEAD,j,Lstag,act,acc_head_half)
                 call mv_end_staging_half(s,TAIL,j,Lstag,act,acc_tail_half)
                 call mv_staging_half(s,j,Lstag,act,acc_bd_half)
              end do
              if (swapping) then
                 where (act) try_swap = try_swap+1
                 call mv_swap(s,Lstag,act,acc_swap,partner,swp)
                 do w=1,NW
                    if (act(w)) call perm_sampling(perm(w),s%isopen(w),s%iworm(w),partner(w),swp(w))
                 end do
              end if
              if (.not. trap) then
                 do w=1,NW
                    if (act(w)) call obdm_accumulate(ep,s%xend(:,:,w),nrho(:,:,w))
                 end do
              end if
           end do
        end if

        end if   ! host-driven / device-resident sampler
        if (trace .and. .not. device_sampler) then       ! PIGS_VPI_TRACE=1: one checksum per walker and step (debugging aid)
           print '(a,2i6,*(1x,z16.16))', ' trace',iblock,istep,(transfer(sum(s%Path(:,:,:,w)),1_8),w=1,NW)
        end if

        ! ---- estimators of the walkers in the diagonal sector (reference vpi.f90:406-473)
        nd = 0
        do w=1,NW
           if (.not. s%isopen(w)) then
              nd = nd+1
              diag_list(nd) = w
              wl(nd) = w-1
           end if
        end do
        if (nd>0) then
           ! LocalEnergy x2 (K4), ThermEnergy (K2/K3) and -- device-resident sampler, PBC -- g(r), S(k) on the device (K7) in
           ! one library call; the host-driven sampler keeps the structural estimators on its mirror.  Accumulated at the top
           ! of the next trip.
           pend_nd = nd
           pend_list(1:nd) = diag_list(1:nd)
           if (device_sampler) then
              pend_struct = .not. trap
              call pigs_check(pigs_diagonal_estimators_begin(ctx,int(nd,c_int32_t),wl,int(Nbin,c_int32_t),rbin, &
                   & int(Nk,c_int32_t),merge(1_c_int32_t,0_c_int32_t,pend_struct)),'pigs_diagonal_estimators_begin')
              est_pending = .true.
           else
              call sampler_flush(s)
              call pigs_check(pigs_diagonal_estimators(ctx,int(nd,c_int32_t),wl,0_c_int32_t,0.d0,0_c_int32_t, &
                   & en9,c_null_ptr,c_null_ptr),'pigs_diagonal_estimators')
              est_have = .true.
           end if
        end if

     end do   ! istep

     ! ---- end of block (reference vpi.f90:477-545)
     if (device_sampler) then
        call pigs_check(pigs_sampler_counters16(ctx,dev_acc),'pigs_sampler_counters16')
        dev_acc0 = dev_acc-dev_acc0
        acc_cm   = int(dev_acc0(1,:));  acc_head  = int(dev_acc0(2,:));  acc_tail = int(dev_acc0(3,:))
        acc_bd   = int(dev_acc0(4,:));  try_open  = int(dev_acc0(5,:));  acc_open = int(dev_acc0(6,:))
        try_close = int(dev_acc0(7,:)); acc_close = int(dev_acc0(8,:));  acc_cm_half = int(dev_acc0(9,:))
        acc_head_half = int(dev_acc0(10,:)); acc_tail_half = int(dev_acc0(11,:)); acc_bd_half = int(dev_acc0(12,:))
        try_swap = int(dev_acc0(13,:)); acc_swap  = int(dev_acc0(14,:)); try_cm   = dble(dev_acc0(15,:))
        try_stag = dble(dev_acc0(16,:))
        dev_acc0 = dev_acc
        if (CWorm>0.d0 .and. .not. trap) then
           ! the device keeps accumulating a walker's OBDM histogram until the block that normalises it
           do w=1,NW
              dev_reset(w) = merge(1,0,idiag_aux(w)/Nstep>=1)
           end do
           call pigs_check(pigs_sampler_nrho(ctx,dev_nrho,dev_reset),'pigs_sampler_nrho')
           nrho = dev_nrho
        end if
     end if
     mE = 0.d0; mT = 0.d0; nd = 0
     vec = 0.d0
     do w=1,NW
        if (idiag_block(w)/=0) then
           BE(:,w)  = BE(:,w)/real(idiag_block(w));  BE2(:,w) = BE2(:,w)/real(idiag_block(w))
           BT(:,w)  = BT(:,w)/real(idiag_block(w));  BT2(:,w) = BT2(:,w)/real(idiag_block(w))
           diag_bl(w) = diag_bl(w)+1
           AE(:,w)  = AE(:,w)+BE(:,w);  AE2(:,w) = AE2(:,w)+BE(:,w)**2
           AT(:,w)  = AT(:,w)+BT(:,w);  AT2(:,w) = AT2(:,w)+BT(:,w)**2
           if (.not. trap) then
              call normalize_gr(ep,density,ngr(w),gr(:,w))
              call normalize_sk(ep,ngr(w),Sk(:,:,w))
              AvGr(:,w) = AvGr(:,w)+gr(:,w); AvGr2(:,w) = AvGr2(:,w)+gr(:,w)*gr(:,w)
              AvSk(:,:,w) = AvSk(:,:,w)+Sk(:,:,w); AvSk2(:,:,w) = AvSk2(:,:,w)+Sk(:,:,w)*Sk(:,:,w)
              vec(21:20+Nbin) = vec(21:20+Nbin)+gr(:,w)
              vec(21+Nbin:20+Nbin+dim*Nk) = vec(21+Nbin:20+Nbin+dim*Nk)+reshape(Sk(:,:,w),[dim*Nk])
              vec(nvec-1) = vec(nvec-1)+1.d0
           end if
           write (ue(w),'(5g20.10e3)') real(iblock),BE(1,w)/Np,BE(2,w)/Np,BE(3,w)/Np
           write (ut(w),'(5g20.10e3)') real(iblock),BT(1,w)/Np,BT(2,w)/Np,BT(3,w)/Np
           write (uh(w),'(i8,6(1x,z16.16))') iblock,BE(1,w)/Np,BE(2,w)/Np,BE(3,w)/Np,BT(1,w)/Np,BT(2,w)/Np,BT(3,w)/Np
           mE = mE+BE(:,w)/Np; mT = mT+BT(:,w)/Np; nd = nd+1
        end if
        if (idiag_aux(w)/Nstep>=1) then
           obdm_bl(w) = obdm_bl(w)+1
           if (.not. trap) then
              call normalize_nr(ep,density,real(idiag_aux(w),8),Nobdm,nrho(:,:,w))
              AvNr(:,:,w) = AvNr(:,:,w)+nrho(:,:,w); AvNr2(:,:,w) = AvNr2(:,:,w)+nrho(:,:,w)*nrho(:,:,w)
              vec(21+Nbin+dim*Nk:20+Nbin+dim*Nk+(Npw+1)*Nbin) = vec(21+Nbin+dim*Nk:20+Nbin+dim*Nk+(Npw+1)*Nbin) &
                   & +reshape(nrho(:,:,w),[(Npw+1)*Nbin])
              vec(nvec) = vec(nvec)+1.d0
           end if
           idiag_aux(w) = 0
           nrho(:,:,w)  = 0.d0
        end if
     end do
     ! ---- the block's one exchange between the shards: all-reduce of the estimator vector
     vec(1) = nd; vec(2:4) = mE; vec(5:7) = mT
     vec(8:20) = [dble(sum(acc_cm)),sum(try_cm),dble(sum(acc_bd)),dble(sum(acc_head)),dble(sum(acc_tail)),sum(try_stag), &
          & dble(sum(idiag_block)),dble(sum(acc_open)),dble(sum(try_open)),dble(sum(acc_close)),dble(sum(try_close)), &
          & dble(sum(acc_swap)),dble(sum(try_swap))]
     if (G>1) call pigs_check(pigs_estimators_allreduce(ctx,vec,int(nvec,c_int32_t)),'pigs_estimators_allreduce')
     ndall = nint(vec(1)); mE = vec(2:4); mT = vec(5:7); cnt_all = vec(8:20)
     ngrall = nint(vec(nvec-1)); nnrall = nint(vec(nvec))
     if (ish==1 .and. NWtot>1) then
        if (ndall>0) then
           write (ueav,'(5g20.10e3)') real(iblock),mE/ndall
           write (utav,'(5g20.10e3)') real(iblock),mT/ndall
        end if
        if (ngrall>0) then                   ! walker average of the block's normalised g(r), S(k)
           ngrav = ngrav+1
           tmp1 = vec(21:20+Nbin)/ngrall
           tmp2 = reshape(vec(21+Nbin:20+Nbin+dim*Nk),[dim,Nk])/ngrall
           AvGrAll = AvGrAll+tmp1; AvGr2All = AvGr2All+tmp1*tmp1
           AvSkAll = AvSkAll+tmp2; AvSk2All = AvSk2All+tmp2*tmp2
        end if
        if (nnrall>0) then
           nnrav = nnrav+1
           tmp3 = reshape(vec(21+Nbin+dim*Nk:20+Nbin+dim*Nk+(Npw+1)*Nbin),[Npw+1,Nbin])/nnrall
           AvNrAll = AvNrAll+tmp3; AvNr2All = AvNr2All+tmp3*tmp3
        end if
     end if
     ! ---- checkpoint (reference vpi.f90:541-545, vpi_mod.f90:263-309): text worldline, particle-major
     if (checkpointing) then
        if (device_sampler) then
           call pigs_check(pigs_path_download_all(ctx,s%Path),'pigs_path_download_all')
           do w=1,NW
              call pigs_check(pigs_sampler_get_rng(ctx,int(w-1,c_int32_t),rpos,s%rng(w)%w),'pigs_sampler_get_rng')
              s%rng(w)%pos = rpos
           end do
           call pigs_check(pigs_sampler_get_worm(ctx,dev_open,dev_iworm,s%xend),'pigs_sampler_get_worm')
           s%isopen = dev_open/=0; s%iworm = dev_iworm
        end if
        do w=1,NW
           suffix = ''
           if (NWtot>1) write (suffix,'(a,i4.4)') '.w',w0+w-1
           open (newunit=ucfg,file='checkpoint'//trim(suffix)//'.dat')
           if (trap) then
              write (ucfg,*) ".True."
           else
              write (ucfg,*) ".False."
           end if
           if (s%isopen(w)) then
              write (ucfg,*) ".True."
           else
              write (ucfg,*) ".False."
           end if
           write (ucfg,*) s%iworm(w)
           do ip=1,Np
              do ib=0,2*Nb
                 write (ucfg,*) (s%Path(k,ip,ib,w),k=1,dim)
              end do
           end do
           write (ucfg,*)
           write (ucfg,*)
           do j=1,2
              write (ucfg,*) (s%xend(k,j,w),k=1,dim)
           end do
           close (ucfg)
           call mt_save(s%rng(w),'rand_state'//trim(suffix))
        end do
     end if
     call system_clock(c1)
     t0 = 0.d0; t1 = dble(c1-c0)/dble(crate)

     if (ish==1) then
     print '(a)',            ' -----------------------------------------------------------'
     print '(a,i8)',         ' BLOCK NUMBER :',iblock
     if (ndall>0) then
        print '(a,3g18.9)',  '   > <E>,<Ec>,<Ep>  =',mE/ndall
        print '(a,3g18.9)',  '   > <Et>,<Kt>,<Vt> =',mT/ndall
     end if
     print '(a,f7.2,a)',     '   > CM movements      =',100*cnt_all(1)/max(cnt_all(2),1.d0),' %'
     print '(a,f7.2,a)',     '   > Staging movements =',100*cnt_all(3)/max(cnt_all(6),1.d0),' %'
     print '(a,f7.2,a)',     '   > Head movements    =',100*cnt_all(4)/max(cnt_all(6),1.d0),' %'
     print '(a,f7.2,a)',     '   > Tail movements    =',100*cnt_all(5)/max(cnt_all(6),1.d0),' %'
     print '(a,f7.2,a)',     '   > Diagonal conf.    =',100.d0*cnt_all(7)/real(Nstep*NWtot),' %'
     print '(a,f7.2,a)',     '   > Open acc          =',100.d0*cnt_all(8)/max(cnt_all(9),1.d0),' %'
     print '(a,f7.2,a)',     '   > Close acc         =',100.d0*cnt_all(10)/max(cnt_all(11),1.d0),' %'
     print '(a,f7.2,a)',     '   > Swap acc          =',100.d0*cnt_all(12)/max(cnt_all(13),1.d0),' %'
     print '(a,f9.2,a,i12,a,i10,a,f9.2,a)', '   > Time per block    =',t1-t0,' s;  Delta S items so far (shard 1)',s%n_eval_items, &
          & ' in',s%n_eval_calls,' batches,',s%t_eval,' s inside them'
     end if

  end do   ! iblock

  !=====================================================================
  ! final averages and files (reference vpi.f90:590-642)
  do w=1,NW
     suffix = ''
     if (NWtot>1) write (suffix,'(a,i4.4)') '.w',w0+w-1
     close (ue(w)); close (ut(w)); close (uh(w))
     if (swapping) then
        open (newunit=k,file='perm_vpi'//trim(suffix)//'.out')
        do ip=1,Np
           write (k,*) ip,perm(w)%histogram(ip)
        end do
        close (k)
     end if
     if (.not. trap) then
        ! written unconditionally, like the reference (a run without a single diagonal / OBDM block
        ! yields NaN columns there too)
        call write_radial('gr_vpi'//trim(suffix)//'.out',ep,diag_bl(w),AvGr(:,w),AvGr2(:,w))
        call write_sk('sk_vpi'//trim(suffix)//'.out',ep,diag_bl(w),AvSk(:,:,w),AvSk2(:,:,w))
        call write_nr('nr_vpi'//trim(suffix)//'.out',ep,obdm_bl(w),AvNr(:,:,w),AvNr2(:,:,w))
     end if
  end do
  if (NWtot>1 .and. ish==1) then
     close (ueav); close (utav)
     if (.not. trap) then                    ! walker-averaged histograms from the reduced block vectors
        call write_radial('gr_vpi.out',ep,ngrav,AvGrAll,AvGr2All)
        call write_sk('sk_vpi.out',ep,ngrav,AvSkAll,AvSk2All)
        call write_nr('nr_vpi.out',ep,nnrav,AvNrAll,AvNr2All)
     end if
  end if

  do w=1,NW
     AE_all(:,w0+w) = AE(:,w); AT_all(:,w0+w) = AT(:,w); diag_bl_all(w0+w) = diag_bl(w)
  end do

  ! final worldlines back from the device must equal the host mirror: the two were kept in
  ! step by commits only
  if (.not. device_sampler) call sampler_flush(s)
  block
    real(8), allocatable :: chk(:,:,:,:)
    allocate (chk(dim,Np,0:2*Nb,NW))
    call pigs_check(pigs_path_download_all(ctx,chk),'pigs_path_download_all')
    if (device_sampler) s%Path = chk
    if (any(chk/=s%Path)) then
       write (0,*) 'pigs_vpi: device worldlines differ from the host mirror'
       stop 3
    end if
    chk_all(:,:,:,w0+1:w0+NW) = chk
  end block

  call sampler_free(s)
  end subroutine run_shard

end program pigs_vpi
