"""Debug helper: host mirror vs device-resident driver through the robust restoration phase, state per iteration."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import madnlp_jl_amd as mj
from madnlp_jl_amd.ipm import MadNLPSolver
from madnlp_jl_amd.ipm_dev import DeviceMadNLPSolver
from madnlp_jl_amd.problems import InfeasibleModel
from test_ipm_restoration_driver import _hip_factory, _options

st = torch.cuda.Stream(); torch.cuda.set_stream(st)
ctx = mj.HipContext(0, stream=st.cuda_stream)
nlp = InfeasibleModel()
h = lambda v: np.asarray(v.cpu().numpy() if hasattr(v, "cpu") else v).ravel()

def make(cls, **kw):
    class D(cls):
        def _record(self, phase=""):
            super()._record(phase)
            RR = getattr(self, "RR", None)
            line = f"k={self.cnt.k} ph={phase or '.'} x={h(self.x)} y={h(self.y)} zl={h(self.zl)} zu={h(self.zu)} c={h(self.c)} f={h(self.f)} jacl={h(self.jacl)} a={self.alpha:.6g} az={self.alpha_z:.6g} dw={self.del_w:.3g}"
            if RR is not None and phase == "R":
                line += f"\n      mu_R={RR.mu_R:.6g} pp={h(RR.pp)} nn={h(RR.nn)} zp={h(RR.zp)} zn={h(RR.zn)} f_R={h(RR.f_R)} objR={RR.obj_val_R:.10g} infs=({RR.inf_pr_R:.6g},{RR.inf_du_R:.6g},{RR.inf_compl_R:.6g})"
            print(line)
        def _rr_finish(self):
            super()._rr_finish()
            RR = self.RR
            print(f"      d: dx={h(self._dx())} dy={h(self._dy())} dzl={h(self._dzl())} dzu={h(self._dzu())} dpp={h(RR.dpp)} dnn={h(RR.dnn)} dzp={h(RR.dzp)} dzn={h(RR.dzn)}")
    return D(nlp, _hip_factory(mj, nlp, ctx), _options(), **kw)

np.set_printoptions(precision=10)
print("==== host"); s = make(MadNLPSolver, sparse=True); s.solve(); print(s.status)
print("==== device"); s = make(DeviceMadNLPSolver); s.solve(); print(s.status)
