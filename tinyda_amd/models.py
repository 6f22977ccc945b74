"""Forward models the device engine can run fused (declared, not opaque Python).

tinyDA's model protocol is "callable theta -> ndarray or (ndarray, qoi)" (posterior.py:95-101).
These classes honour it on the host and additionally expose what the HIP kernels need.
"""
import numpy as np


class LinearModel:
    """F(theta) = A theta (+ b).  All BASELINE.json configurations use linear forward models."""

    def __init__(self, A, b=None):
        self.A = np.ascontiguousarray(np.asarray(A, dtype=np.float64))
        if self.A.ndim != 2:
            raise ValueError("A must be a 2-D array (observations x parameters)")
        self.b = None if b is None else np.ascontiguousarray(np.asarray(b, dtype=np.float64))
        if self.b is not None and self.b.shape != (self.A.shape[0],):
            raise ValueError("b must have one entry per observation")

    def __call__(self, parameters):
        out = self.A @ np.asarray(parameters, dtype=np.float64)
        return out if self.b is None else out + self.b

    def gradient(self, parameters, sensitivity):
        """J^T s, the method MALA looks for on a model (proposal.py:938-943, :996-998)."""
        return self.A.T @ np.asarray(sensitivity, dtype=np.float64)


class Rosenbrock:
    """The reference's Rosenbrock example (examples/MALA Rosenbrock.ipynb) as a d-parameter chain with one
    scalar output: F(theta) = [ sum_i (a - theta_i)^2 + b (theta_{i+1} - theta_i^2)^2 ].  With data [0] and unit
    variance the log-likelihood is -F^2/2 (BASELINE config 4 uses d = 32, a = 1, b = 10)."""

    def __init__(self, a=1.0, b=10.0):
        self.a, self.b = float(a), float(b)

    def __call__(self, parameters):
        t = np.asarray(parameters, dtype=np.float64)
        return np.array([np.sum((self.a - t[:-1]) ** 2 + self.b * (t[1:] - t[:-1] ** 2) ** 2)])


class DeviceModel:
    """A (possibly non-linear) forward model given as HIP source, compiled at run time into the fused step kernel
    (extension; tinyDA only knows Python callables).  The source must define

        __device__ double tda_forward(const double* theta, int dim, int o);   // output o of F(theta), 0 <= o < n_outputs

    `reference`, if given, is a Python callable theta -> outputs used when the model is called on the host (host
    protocol, tests); without it the model only runs on the device."""

    def __init__(self, source, n_outputs, reference=None):
        self.source = str(source)
        self.n_outputs = int(n_outputs)
        self.reference = reference
        if "tda_forward" not in self.source:
            raise ValueError("the source must define __device__ double tda_forward(const double* theta, int dim, int o)")

    def __call__(self, parameters):
        if self.reference is None:
            raise TypeError("this DeviceModel has no host reference implementation; run it with backend='hip'")
        return np.atleast_1d(np.asarray(self.reference(np.asarray(parameters, dtype=np.float64)), dtype=np.float64))


class BatchedModel:
    """An arbitrary host forward model evaluated for ALL chains at once (extension; tinyDA calls the model once per chain
    and step, posterior.py:95-96).  `fn` maps an (n_chains, dim) array of parameters to an (n_chains, n_outputs) array of
    model outputs -- a vectorised NumPy function, a batched solver, a remote service.  On the device path the engine hands
    it each step's proposals in one call and keeps proposals, log-densities, accept test, adaptation and records on the
    GPU; called with a single parameter vector it honours the reference's model protocol.

    inplace=True: `fn(parameters, out)` writes the outputs into `out` (the engine's page-locked staging buffer) instead
    of returning a new array."""

    def __init__(self, fn, n_outputs, inplace=False):
        self.fn = fn
        self.n_outputs = int(n_outputs)
        self.inplace = bool(inplace)

    def batch(self, parameters, out=None):
        parameters = np.asarray(parameters, dtype=np.float64)
        if not self.inplace:
            res = np.asarray(self.fn(parameters), dtype=np.float64)
            if out is None:
                return res
            if res.shape != out.shape:
                raise ValueError("the batched model returned shape %s, expected %s" % (res.shape, out.shape))
            out[...] = res
            return out
        if out is None:
            out = np.empty((parameters.shape[0], self.n_outputs))
        self.fn(parameters, out)
        return out

    single = None  # optional one-vector form of the same model (set when a plain callable was wrapped)

    def __call__(self, parameters):
        if self.single is not None:
            return self.single(parameters)
        return self.batch(np.asarray(parameters, dtype=np.float64)[None, :])[0]
