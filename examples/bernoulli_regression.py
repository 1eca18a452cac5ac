"""The flow of the reference's examples/bernoulli_regression.py (no plotting): one sparse Bernoulli regression started from
the complement of the true adjacency.

    python examples/bernoulli_regression.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyglm_amd.regression import SparseBernoulliRegression

N, B, T = 2, 1, 1000
true_reg = SparseBernoulliRegression(N, B)
X = np.random.randn(T, N * B)
y = true_reg.rvs(X=X)

test_reg = SparseBernoulliRegression(N, B)
test_reg.a = np.bitwise_not(true_reg.a)

As, lps = [], []
for _ in range(100):
    test_reg.resample([(X, y)])
    As.append(test_reg.a.copy())
    lps.append(test_reg.log_likelihood((X, y)).sum())

print("True A: {}".format(true_reg.a))
print("Mean A: {}".format(np.mean(As, axis=0)))
print("True W: {}   last W: {}".format(true_reg.W.ravel(), test_reg.W.ravel()))
print("log likelihood: first %.1f  last %.1f" % (lps[0], lps[-1]))
