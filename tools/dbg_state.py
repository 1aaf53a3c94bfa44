import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
os.environ["AQ_TT"] = "2"; os.environ["AQ_STAGGER"] = "0"; os.environ["AQ_CHAIN"] = "0"
import atlasqtl_amd as A
from oracle import atlasqtl_oracle as O
from tests.util import make_problem
n, p, q = 1000, 208, 40
prob = make_problem(n, p, q, p_act=8, prob_assoc=0.3)
for nsw in (1, 2):
    ref = O.atlasqtl_global_local_core_(prob["Y"], prob["X"], q, None, 1, 0.1, nsw, prob["list_hyper"], prob["list_init"], full_output=True)
    got = A.atlasqtl_global_local_core_(prob["Y"], prob["X"], q, None, 1, 0.1, nsw, 0, prob["list_hyper"], prob["list_init"], full_output=True, debug=False)
    dg = np.abs(got["gam_vb"] - ref["gam_vb"]); dm = np.abs(got["mu_beta_vb"] - ref["mu_beta_vb"])
    print("sweeps", nsw, "max dgam", dg.max(), "max dmu", dm.max(), "tau rel", np.max(np.abs(got["tau_vb"]/ref["tau_vb"]-1)), "lb", got["lb_opt"], ref["lb_opt"])
    pp = dg.shape[0]
    print(" per SNP block max dmu:", [float(f"{dm[16*b:16*b+16].max():.1e}") for b in range((pp+15)//16)])
    print(" per trait max dmu:", [float(f"{dm[:,k].max():.1e}") for k in range(q)])
    print(" tau rel per trait:", [float(f"{abs(got['tau_vb'][k]/ref['tau_vb'][k]-1):.1e}") for k in range(q)])
