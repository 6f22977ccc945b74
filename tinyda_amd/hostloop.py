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


class _Rung:
    """One level of a hierarchy on the host: its links, the accept flag of every entry and whether the entry is a step
    the level made itself (False for the initial link and for the entries alignment appends)."""

    __slots__ = ("posterior", "links", "took", "own")

    def __init__(self, posterior, first_link):
        self.posterior = posterior
        self.links = [first_link]
        self.took = [True]
        self.own = [False]

    def push(self, link, took, own):
        self.links.append(link)
        self.took.append(took)
        self.own.append(own)

    def local_links(self):
        return [ln for ln, mine in zip(self.links, self.own) if mine]


class HierarchyChain:
    """Delayed Acceptance (two posteriors; the rules of tinyDA/chain.py:185-530) and Multilevel Delayed Acceptance (three or
    more; chain.py:570-769 with the level logic of proposal.py:1285-1624) for ONE chain on the host, for everything the
    device engine does not lower: opaque models (also returning (output, qoi)), any object speaking the proposal protocol
    at the base level (DREAMZ as in the reference's MLDA notebook, MALA, user classes), more than four levels, more than 64
    parameters.  One flat list of rungs and one routine per kind of step instead of the reference's nested proposal
    objects; every random number is asked for in the reference's order (checked by replaying its traces,
    tests/test_host_hierarchy.py)."""

    def __init__(self, posteriors, proposal, subchain_lengths, initial_parameters=None, adaptive_error_model=None,
                 store_coarse_chain=True, randomize_subchain_length=False, error_model_covariance="dense"):
        from .moments import RecursiveSampleMoments, ZeroMeanRecursiveSampleMoments

        if error_model_covariance not in ("dense", "diagonal"):
            raise ValueError("error_model_covariance must be 'dense' (the reference) or 'diagonal'")
        self.diagonal_bias = error_model_covariance == "diagonal"  # extension: only the diagonal of a bias covariance is used

        if adaptive_error_model not in (None, "state-independent", "state-dependent"):
            raise ValueError("Adaptive error model can only be state-dependent, state-independent or None.")
        n = len(posteriors)
        if n < 2 or len(subchain_lengths) != n - 1:
            raise ValueError("a hierarchy needs at least two posteriors and one subchain length per pair of levels")
        self.two_level = n == 2
        self.lengths = [int(v) for v in subchain_lengths]
        self.proposal = proposal
        self.error_model = adaptive_error_model
        self.keep_coarse = bool(store_coarse_chain)
        self.randomize = bool(randomize_subchain_length)
        if self.randomize:
            if not self.two_level:
                raise NotImplementedError("randomize_subchain_length is a Delayed Acceptance (two-level) option")
            if self.lengths[0] == 1:
                raise ValueError("Randomize subchain length requires a subchain_length > 1.")
            if not self.keep_coarse:
                raise ValueError("Randomize subchain length requires storing the coarse chain.")
        if initial_parameters is None:  # chain.py:251 draws from the coarse prior, :631 from the finest
            initial_parameters = posteriors[0 if self.two_level else -1].prior.rvs()
        self.initial_parameters = initial_parameters
        # first links in the reference's order of model calls: coarse then fine for DA, finest first for MLDA
        first = {}
        for k in (range(n) if self.two_level else reversed(range(n))):
            first[k] = posteriors[k].create_link(initial_parameters)
        self.rungs = [_Rung(posteriors[k], first[k]) for k in range(n)]
        proposal.setup_proposal(parameters=initial_parameters, posterior=posteriors[0])
        self.promoted = []
        self.effective_lengths = []
        # ---- error model (chain.py:268-305; proposal.py:1404-1417, 1442-1467) ----
        self.trackers = [None] * n  # trackers[q]: moments of F_q - F_{q-1}
        self.last_diff = [None] * n
        if self.error_model is None:
            return
        if not self.two_level and self.error_model == "state-dependent":
            raise NotImplementedError("the state-dependent error model is a Delayed Acceptance (two-level) option")
        for q in range(1, n):
            diff = self.rungs[q].links[-1].model_output - self.rungs[q - 1].links[-1].model_output
            self.last_diff[q] = diff
            m = diff.shape[0]
            if self.error_model == "state-independent":
                self.trackers[q] = RecursiveSampleMoments(diff, np.zeros((m, m)))
            else:
                self.trackers[q] = ZeroMeanRecursiveSampleMoments(np.zeros((m, m)))
        for q in reversed(range(1, n)):
            self._install_bias(q)

    # ---- variates: one method per kind so that a test can feed recorded ones --------------------------------
    def _uniform(self, level):
        return np.random.random()

    def _promoted_index(self, length):
        return np.random.randint(-length, 0)

    # ---- error model ----------------------------------------------------------------------------------------
    def _install_bias(self, q):
        """hand level q - 1 the bias it is corrected with and re-evaluate its latest link (posterior.update_link)"""
        below = self.rungs[q - 1]
        like = below.posterior.likelihood
        shape = (lambda sig: np.diag(np.diag(sig))) if self.diagonal_bias else (lambda sig: sig)
        if self.error_model == "state-dependent":
            like.set_bias(self.last_diff[q], shape(self.trackers[q].get_sigma()))
        elif self.two_level or q == len(self.rungs) - 1:
            like.set_bias(self.trackers[q].get_mu(), shape(self.trackers[q].get_sigma()))
        else:  # biases stack upwards: level q - 1 sees the sum of every tracker from q to the finest (proposal.py:1563-1569)
            stack = self.trackers[q:]
            like.set_bias(np.sum([t.get_mu() for t in stack], axis=0), shape(np.sum([t.get_sigma() for t in stack], axis=0)))
        below.links[-1] = below.posterior.update_link(below.links[-1])

    def _learn_bias(self, q):
        """after a step of level q: feed the tracker of the pair (q - 1, q) and refresh level q - 1"""
        me, below = self.rungs[q], self.rungs[q - 1]
        if self.two_level:  # Delayed Acceptance: always the current pair (chain.py:485-523)
            now = me.links[-1].model_output - below.links[-1].model_output
            if self.error_model == "state-independent":
                self.trackers[q].update(now)
            else:
                self.trackers[q].update(me.links[-1].model_output - (below.links[-1].model_output + self.last_diff[q]))
            self.last_diff[q] = now
        else:  # MLDA: the difference is refreshed on acceptance only, the stale one is fed again otherwise (chain.py:742-753)
            if me.took[-1]:
                self.last_diff[q] = me.links[-1].model_output - below.links[-1].model_output
            self.trackers[q].update(self.last_diff[q])
        self._install_bias(q)

    # ---- steps ----------------------------------------------------------------------------------------------
    def _base_step(self):
        rung = self.rungs[0]
        here = rung.links[-1]
        there = rung.posterior.create_link(self.proposal.make_proposal(here))
        alpha = self.proposal.get_acceptance(there, here)
        moved = bool(self._uniform(0) < alpha)
        rung.push(there if moved else here, moved, True)
        self.proposal.adapt(parameters=rung.links[-1].parameters, parameters_previous=rung.links[-2].parameters,
                            accepted=rung.took)

    def _realign(self, upto, parameters, took):
        """every level below `upto` re-appends its latest link at `parameters` (identity, like proposal.py:1469-1490)"""
        for j in reversed(range(upto)):
            rung = self.rungs[j]
            match = next(ln for ln in reversed(rung.links) if ln.parameters is parameters)
            rung.push(match, took, False)

    def _upper_step(self, q):
        """one step of level q >= 1 under the multilevel rules (proposal.py:1507-1581, chain.py:697-765)"""
        me, below, length = self.rungs[q], self.rungs[q - 1], self.lengths[q - 1]
        for _ in range(length):
            self._step(q - 1)
        top = q == len(self.rungs) - 1
        if sum(below.took[-length:]) == 0:  # nothing moved below: no model call, a recorded rejection
            me.push(me.links[-1], False, not top)
        else:
            there = me.posterior.create_link(below.links[-1].parameters)
            alpha = np.exp(there.posterior - me.links[-1].posterior
                           + below.links[-(length + 1)].posterior - below.links[-1].posterior)
            moved = bool(self._uniform(q) < alpha)
            me.push(there if moved else me.links[-1], moved, not top)
        self._realign(q, me.links[-1].parameters, me.took[-1])
        if self.error_model is not None:
            self._learn_bias(q)

    def _fine_step(self):
        """one fine step of Delayed Acceptance (chain.py:342-402)"""
        coarse, fine, length = self.rungs[0], self.rungs[1], self.lengths[0]
        if not self.keep_coarse:
            coarse.links = [coarse.links[-1]]
        for _ in range(length):
            self._base_step()
        start = coarse.links[-(length + 1)]
        if sum(coarse.took[-length:]) == 0:
            fine.push(fine.links[-1], False, False)
            coarse.push(start, False, False)
        else:
            index = self._promoted_index(length) if self.randomize else -1
            pick = coarse.links[index]
            there = fine.posterior.create_link(pick.parameters)
            self.promoted.append(pick)
            self.effective_lengths.append(index + length + 1)
            here = fine.links[-1]
            if self.error_model == "state-dependent":  # chain.py:446-473
                shifted = coarse.posterior.update_link(start, there.model_output - pick.model_output)
                if self.proposal.is_symmetric:
                    q_xy = q_yx = 0
                else:
                    q_xy = self.proposal.get_q(here, there)
                    q_yx = self.proposal.get_q(there, here)
                alpha = np.exp(min(there.posterior + q_yx, shifted.posterior + q_xy)
                               - min(here.posterior + q_xy, pick.posterior + q_yx))
            else:  # chain.py:475-483
                alpha = np.exp(there.posterior - here.posterior + start.posterior - pick.posterior)
            if self._uniform(1) < alpha:
                fine.push(there, True, False)
                coarse.push(pick, True, False)
            else:
                fine.push(here, False, False)
                coarse.push(start, False, False)
        if self.error_model is not None:
            self._learn_bias(1)

    def _step(self, level):
        if level == 0:
            self._base_step()
        else:
            self._upper_step(level)

    def sample(self, iterations, progressbar=False):
        steps = range(iterations)
        if progressbar:
            try:
                from tqdm import tqdm

                steps = tqdm(steps)
            except ImportError:
                pass
        top = len(self.rungs) - 1
        for _ in steps:
            if self.two_level:
                self._fine_step()
            else:
                if not self.keep_coarse:  # MLDA._reset_chain (proposal.py:1492-1497)
                    for rung in self.rungs[:-1]:
                        rung.links = [rung.links[-1]]
                self._upper_step(top)

    # ---- results --------------------------------------------------------------------------------------------
    def level_chain(self, level):
        """what sample() returns for a level: every link of the finest chain, the level's own steps below it
        (sampler.py:421-427, :535-538); None for coarse levels that were not stored"""
        if level == len(self.rungs) - 1:
            return self.rungs[level].links
        return self.rungs[level].local_links() if self.keep_coarse else None
