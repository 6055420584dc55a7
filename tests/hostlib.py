"""Build + load the Fortran host library for HOST-LOGIC tests on a machine without a GPU:
it is linked against tests/shim (the C ABI implemented with the CPU oracle -- test
infrastructure only).  On the GPU box the same host sources link against libpigs_hip.so."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "pathintegralgroundstate_amd", "host")
BUILD = os.path.join(ROOT, "tests", "shim", "_build")
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


def build_cpu_host():
    os.makedirs(BUILD, exist_ok=True)
    shim = os.path.join(BUILD, "libpigs_cpu_shim.so")
    srcs = [os.path.join(ROOT, "tests", "shim", "pigs_cpu_shim.c"), os.path.join(ROOT, "oracle", "pigs_oracle.c")]
    if not os.path.exists(shim) or any(os.path.getmtime(s) > os.path.getmtime(shim) for s in srcs):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-ffp-contract=off", "-shared", "-pthread", "-o", shim] + srcs + ["-lm"])
    subprocess.check_call(["make", "-s", "-B", "-C", HOST, f"OUT={BUILD}", "OBJ=/tmp/pigs_host_obj_cpu",
                           f"BACKEND_DIR={BUILD}", "BACKEND=pigs_cpu_shim", "all"])
    return shim, os.path.join(BUILD, "libpigs_host.so"), os.path.join(BUILD, "pigs_vpi")


class PigsParams(C.Structure):
    _fields_ = [("dim", C.c_int32), ("Np", C.c_int32), ("Nb", C.c_int32), ("Nmax", C.c_int32),
                ("trap", C.c_int32), ("wf_table", C.c_int32), ("v_table", C.c_int32), ("reserved", C.c_int32),
                ("dr", C.c_double), ("rcut2", C.c_double), ("dt", C.c_double), ("Rm", C.c_double),
                ("Lbox", C.c_double * 3), ("a_ho", C.c_double * 3)]


MOVES = {"TranslateChain": 1, "Bisection": 2, "MoveHeadBisection": 3, "MoveTailBisection": 4, "Staging": 5,
         "MoveHead": 6, "MoveTail": 7, "TranslateHalfChain": 8, "StagingHalfChain": 9,
         "MoveHeadHalfChain": 10, "MoveTailHalfChain": 11, "OpenChain": 12, "CloseChain": 13, "Swap": 14}


class HostSampler:
    """W lock-step walkers of the Fortran host sampler over a C-ABI backend."""

    def __init__(self, S, VT, WF, W=1, backend=None, hostlib=None):
        if backend is None:
            backend, hostlib, _ = build_cpu_host()
        self.B = C.CDLL(backend, mode=C.RTLD_GLOBAL)
        self.H = C.CDLL(hostlib)
        self.S, self.W = S, W
        p = PigsParams()
        p.dim, p.Np, p.Nb, p.Nmax = S.dim, S.Np, S.Nb, S.Nmax
        p.trap, p.wf_table, p.v_table = int(S.trap), 1, 1
        p.dr, p.rcut2, p.dt, p.Rm = S.dr, S.rcut2, S.dt, S.Rm
        for k in range(3):
            p.Lbox[k] = S.Lbox[k]
            p.a_ho[k] = S.a_ho[k]
        self._VT, self._WF = np.ascontiguousarray(VT, float), np.ascontiguousarray(WF, float)
        self.ctx = C.c_void_p()
        self.B.pigs_ctx_create.argtypes = [C.POINTER(PigsParams), _dp, _dp, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
        rc = self.B.pigs_ctx_create(C.byref(p), self._VT.ctypes.data_as(_dp), self._WF.ctypes.data_as(_dp), W, 0,
                                    C.byref(self.ctx))
        assert rc == 0
        H = self.H
        H.hs_create.argtypes = [C.c_int] * 5 + [C.c_double] * 3 + [_dp, C.c_void_p]
        H.hs_create.restype = C.c_int
        H.hs_set_path.argtypes = [C.c_int, C.c_int, _dp]
        H.hs_get_path.argtypes = [C.c_int, C.c_int, _dp]
        H.hs_set_rng.argtypes = [C.c_int, C.c_int, C.c_int, _ip]
        H.hs_get_rng.argtypes = [C.c_int, C.c_int, _ip, _ip]
        H.hs_set_worm.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, _dp]
        H.hs_get_worm.argtypes = [C.c_int, C.c_int, _ip, _ip, _dp]
        H.hs_move.argtypes = [C.c_int] * 4 + [C.c_double, _ip, _ip, _ip, _ip, _ip]
        H.hs_uniform_stream.argtypes = [C.c_int, C.c_int, _dp]
        H.hs_gauss_stream.argtypes = [C.c_int, C.c_int, _dp]
        Lb = np.ascontiguousarray(S.Lbox[:S.dim], float)
        self.h = H.hs_create(S.dim, S.Np, S.Nb, W, int(S.trap), S.dt, S.density, S.CWorm, Lb.ctypes.data_as(_dp), self.ctx)
        assert self.h > 0

    def close(self):
        if getattr(self, "h", 0) > 0:
            self.H.hs_destroy(self.h)
            self.B.pigs_ctx_destroy.argtypes = [C.c_void_p]
            self.B.pigs_ctx_destroy(self.ctx)
            self.h = 0

    def set_path(self, w, P):
        P = np.ascontiguousarray(P, float)
        self.H.hs_set_path(self.h, w, P.ctypes.data_as(_dp))

    def get_path(self, w):
        P = np.empty((self.S.M, self.S.Np, self.S.dim))
        self.H.hs_get_path(self.h, w, P.ctypes.data_as(_dp))
        return P

    def upload(self):
        self.H.hs_upload(self.h)

    def device_paths(self):
        self.H.hs_flush(self.h)
        out = np.empty((self.W, self.S.M, self.S.Np, self.S.dim))
        self.B.pigs_path_download_all.argtypes = [C.c_void_p, _dp]
        self.B.pigs_path_download_all(self.ctx, out.ctypes.data_as(_dp))
        return out

    def seed(self, w, seed):
        self.H.hs_seed(self.h, w, seed)

    def set_rng(self, w, mti, mt):
        mt = np.ascontiguousarray(mt, np.uint32).view(np.int32)
        self.H.hs_set_rng(self.h, w, mti, mt.ctypes.data_as(_ip))

    def get_rng(self, w):
        pos = C.c_int32()
        mt = np.zeros(624, np.int32)
        self.H.hs_get_rng(self.h, w, C.byref(pos), mt.ctypes.data_as(_ip))
        return pos.value, mt.view(np.uint32).copy()

    def set_worm(self, w, isopen, iworm, xend):
        xend = np.ascontiguousarray(xend, float)
        self.H.hs_set_worm(self.h, w, int(isopen), iworm, xend.ctypes.data_as(_dp))

    def get_worm(self, w):
        o, i = C.c_int32(), C.c_int32()
        xe = np.zeros((2, self.S.dim))
        self.H.hs_get_worm(self.h, w, C.byref(o), C.byref(i), xe.ctypes.data_as(_dp))
        return bool(o.value), i.value, xe

    def move(self, name, ip_of, i1=0, i2=0, rpar=0.0, active=None, accepted=None):
        W = self.W
        ipo = np.ascontiguousarray(np.broadcast_to(ip_of, (W,)), np.int32)
        act = np.ones(W, np.int32) if active is None else np.ascontiguousarray(active, np.int32)
        acc = np.zeros(W, np.int32) if accepted is None else np.ascontiguousarray(accepted, np.int32)
        par, swp = np.zeros(W, np.int32), np.zeros(W, np.int32)
        self.H.hs_move(self.h, MOVES[name], i1, i2, rpar, ipo.ctypes.data_as(_ip), act.ctypes.data_as(_ip),
                       acc.ctypes.data_as(_ip), par.ctypes.data_as(_ip), swp.ctypes.data_as(_ip))
        return acc, par, swp
