import sys
sys.path.insert(0, "/root/repo")
from tests.util import make_problem
import atlasqtl_amd as A
prob = make_problem(100, 75, 20, p_act=10, prob_assoc=1.0)
got = A.atlasqtl_global_local_core_(prob["Y"], prob["X"], 20, None, 1, 0.1, 50, 0, prob["list_hyper"], prob["list_init"])
import torch
print("torch after lib:", torch.cuda.is_available(), torch.empty(4, device="cuda").sum().item())
