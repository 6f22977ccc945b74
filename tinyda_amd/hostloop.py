"""Host chain driver for posteriors the device engine cannot lower (opaque Python forward models).

Same per-step protocol as tinyDA/chain.py:37-129; this is BASELINE config 1's "plumbing, no GPU" path.
Device-lowerable problems never come here: `sample()` sends them to the HIP engine, and raises if the
engine is unavailable instead of silently computing on the CPU.
"""
import numpy as np


class Chain:
    def __init__(self, posterior, proposal, initial_parameters=None):
        self.posterior = posterior
        self.proposal = proposal
        self.initial_parameters = posterior.prior.rvs() if initial_parameters is None else initial_parameters
        self.chain = [posterior.create_link(self.initial_parameters)]
        self.accepted = [True]
        proposal.setup_proposal(parameters=self.initial_parameters, posterior=posterior)

    def sample(self, iterations, progressbar=True):
        steps = range(iterations)
        bar = None
        if progressbar:
            try:
                from tqdm import tqdm

                bar = steps = tqdm(steps)
            except ImportError:
                bar = None
        for _ in steps:
            if bar is not None:
                bar.set_description("Running chain, α = %0.2f" % np.mean(self.accepted[-100:]))
            current = self.chain[-1]
            candidate = self.posterior.create_link(self.proposal.make_proposal(current))
            alpha = self.proposal.get_acceptance(candidate, current)
            took = bool(np.random.random() < alpha)
            self.chain.append(candidate if took else current)
            self.accepted.append(took)
            self.proposal.adapt(parameters=self.chain[-1].parameters,
                                parameters_previous=self.chain[-2].parameters, accepted=self.accepted)
        if bar is not None:
            bar.close()
