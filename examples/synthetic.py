"""The flow of the reference's examples/synthetic.py on the MI355X path (no plotting): simulate a network with
self-inhibition, fit a SparseBernoulliGLM by Gibbs sampling, print the log-likelihood trace and the posterior means.

    python examples/synthetic.py [N_samples]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
np.random.seed(0)

from pyglm_amd.utils.basis import cosine_basis
from pyglm_amd.models import SparseBernoulliGLM

T = 10000   # time bins
N = 4       # neurons
B = 1       # basis functions
L = 100     # autoregressive window

basis = cosine_basis(B=B, L=L) / L

true_model = SparseBernoulliGLM(N, basis=basis, regression_kwargs=dict(S_w=10.0, mu_b=-2.))
for n in range(N):
    true_model.regressions[n].a[n] = True
    true_model.regressions[n].W[n, :] = -2.0
_, Y = true_model.generate(T=T, keep=True)

test_model = SparseBernoulliGLM(N, basis=basis, regression_kwargs=dict(S_w=10.0, mu_b=-2.))
test_model.add_data(Y)

N_samples = int(sys.argv[1]) if len(sys.argv) > 1 else 100
lps, W_smpls, A_smpls, b_smpls = [], [], [], []
for itr in range(N_samples):
    test_model.resample_model()
    lps.append(test_model.log_likelihood())
    W_smpls.append(test_model.weights.copy())
    A_smpls.append(test_model.adjacency.copy())
    b_smpls.append(test_model.biases.copy())
    if itr % 10 == 0:
        print("iteration %3d  log likelihood %.1f" % (itr, lps[-1]))

half = N_samples // 2
print("true log likelihood      %.1f" % true_model.log_likelihood())
print("mean log likelihood      %.1f" % np.mean(lps[half:]))
print("posterior mean adjacency\n", np.mean(A_smpls[half:], axis=0).round(2))
print("posterior mean weights\n", np.mean(W_smpls[half:], axis=0)[:, :, 0].round(2))
print("posterior mean biases   ", np.mean(b_smpls[half:], axis=0).round(2))
