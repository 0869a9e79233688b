"""MI355X-native Kalman filter / RTS smoother engine behind TemporalGPs.jl's LGSSM interface.

Import as `import temporalgps_jl_amd as tgp` (the directory name contains a dot, so the repo root ships a
one-file import shim `temporalgps_jl_amd.py`).
"""
from . import _lib
from .lgssm import (LGSSM, PosteriorLGSSM, Forward, Gaussian, GaussMarkovModel, Reverse, ScalarOutputLGC, SmallOutputLGC, LargeOutputLGC, BottleneckLGC, _filter, logpdf, marginals,
                    posterior, posterior_marginals, logpdf_and_posterior_marginals, posterior_marginals_at, rand, replace_observation_noise_cov)

from .multi import MultiLGSSM

__all__ = ["MultiLGSSM", "LGSSM", "PosteriorLGSSM", "Forward", "Reverse", "Gaussian", "GaussMarkovModel", "ScalarOutputLGC", "SmallOutputLGC", "LargeOutputLGC", "BottleneckLGC", "logpdf", "_filter",
           "posterior", "marginals", "posterior_marginals", "logpdf_and_posterior_marginals", "posterior_marginals_at", "rand", "replace_observation_noise_cov", "_lib"]
