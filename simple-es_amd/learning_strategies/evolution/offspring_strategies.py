"""simple_genetic / simple_evolution / openai_es with the reference's constructor signatures and methods
(learning_strategies/evolution/offspring_strategies.py:11-434), re-designed around a device-resident
population:

  reference                                    here
  -----------------------------------------    -----------------------------------------------------------
  list of N {agent_id: torch module}           Population: float32[N_local, P] on the GPU (+ lazy module view)
  deepcopy + np.random.normal per offspring    one Philox perturbation kernel over the shard (ses_perturb)
  python argsort / rank loop / z-score         ses_rank_center (exact counting rank, float64 weights)
  python sum over N modules + numpy Adam       ses_es_update_* (ES gradient + Adam fused)
  in-place elite sum over modules              ses_elite_ids / ses_perturb(row_ids) / ses_elite_mean

Population layouts and quirks are kept: openai_es has N members with member 0 = mu;
simple_evolution has N+1 = [mu, elite0, N-1 children] and its elite "mean" reproduces the reference's
in-place aliasing; simple_genetic has k*(N//k) members and decays sigma AFTER regenerating.

noise="philox" (default): counter-based device noise; any GPU count gives the same population.
noise="numpy": the reference's own stream -- float64 draws from the global numpy generator, uploaded and
applied on the device with the reference's float64->float32 rounding; with the same seed the populations
and parent updates are bit-identical to the reference (single process only).
"""
import numpy as np
import torch

from learning_strategies.optimizers import Adam
from ses import HipES
from ses.parallel import Shard, attach_comm

from .abstracts import BaseOffspringStrategy
from .utils import wrap_agentid


class Population:
    """The offspring group handed from a strategy to ESLoop.  Device view: `theta` (this rank's rows);
    sequence view (len / index / iterate) materialises {agent_id: module} dicts like the reference's list."""

    def __init__(self, theta, shard, network, agent_ids, gen):
        self.theta = theta            # float32[n_local, P] on the device
        self.shard = shard
        self.gen = gen                # generation counter (keys the env-init Philox stream in ESLoop)
        self._network = network
        self._agent_ids = agent_ids

    def __len__(self):
        return self.shard.n_global

    def __getitem__(self, i):
        if not 0 <= i < self.shard.n_global:
            raise IndexError(i)
        j = i - self.shard.first
        if not 0 <= j < self.shard.n_local:
            raise IndexError(f"offspring {i} lives on another rank")
        import copy
        net = copy.deepcopy(self._network).load_flat(self.theta[j].cpu().numpy())
        return wrap_agentid(self._agent_ids, net)

    def __iter__(self):
        return (self[i] for i in range(self.shard.first, self.shard.first + self.shard.n_local))


class PendingReward:
    """best_reward of a generation, read back without stalling the thread that enqueues the next one and without an
    event in the launch stream: the kernel that finds the maximum stores it straight into pinned host memory, which
    the host pre-set to NaN when it enqueued the generation; result() polls until the value is there (falling back to
    a stream synchronisation if it does not appear -- a NaN maximum would mean NaN returns)."""

    def __init__(self, host):
        self._host, self._value = host, None

    def result(self):
        if self._value is None:
            v = float(self._host[0])
            spins = 0
            while v != v:
                spins += 1
                if spins > 2000000:                     # ~seconds: give up polling, wait for the device
                    torch.cuda.synchronize()
                    v = float(self._host[0])
                    break
                v = float(self._host[0])
            self._value = v
        return self._value


class _ReadbackRing:
    """`depth` pinned host floats reused round-robin (the loop keeps at most two generations in flight).  `best` is the
    slot the current generation's kernels write -- host memory the GPU can store to, so neither a copy nor an event sits
    in the launch stream (a queued 4-byte device-to-host copy cost ~10 us of stream time per generation, an event ~5)."""

    def __init__(self, device, depth=4):
        self.slots = [torch.full((1,), float("nan"), dtype=torch.float32).pin_memory() for _ in range(depth)]
        self.k = 0

    @property
    def best(self):
        return self.slots[self.k]

    def arm(self):
        """call BEFORE enqueueing the kernels that write `best`: marks the slot as pending"""
        self.slots[self.k][0] = float("nan")
        return self.slots[self.k]

    def push(self):
        host = self.slots[self.k]
        self.k = (self.k + 1) % len(self.slots)
        return PendingReward(host)


class _DeviceStrategy(BaseOffspringStrategy):
    def __init__(self, init_sigma, sigma_decay, offspring_num, noise, seed):
        if noise not in ("philox", "numpy"):
            raise ValueError("noise must be 'philox' or 'numpy'")
        self.offspring_num = offspring_num
        self.init_sigma = init_sigma
        self.sigma_decay = sigma_decay
        self.curr_sigma = init_sigma
        self.noise = noise
        self.seed = int(seed)
        self.gen = 0                  # bumped by every _gen_offsprings: Philox generation key
        self.dev = None
        self._elite_ids_dev = None

    # ---- plumbing -----------------------------------------------------------------------------
    def _bind(self, network, agent_ids):
        self.agent_ids = agent_ids
        self.network = network
        network.zero_init()           # every strategy starts from the zero network (offspring_strategies.py:83,200,348)
        self.dev = HipES(None, network.num_state, network.num_action, network.discrete_action, network.use_gru)
        # multi-GPU: the shard form of the openai_es tail exchanges its chunk partials over the transports ESLoop attached
        # (attach_comm there is the collective one).  Here: reuse only -- a strategy built on some ranks never starts a rendezvous
        attach_comm(self.dev, create=False)
        self.P = self.dev.P
        self._ring = _ReadbackRing(self.dev.device)
        self._shards = {}

    def _shard(self, n):
        sh = self._shards.get(n)
        if sh is None:
            sh = self._shards[n] = Shard(n)
        return sh

    def _const_map(self, key, build):
        """The parent map of a strategy depends only on (elite_num, offspring_num): built once, so that the identity
        checks of the upload caches below hit every generation."""
        maps = self.__dict__.setdefault("_maps", {})
        if key not in maps:
            maps[key] = build()
        return maps[key]

    def evaluate(self, rewards):
        """(offspring_group, best_reward, curr_sigma) like the reference; best_reward is read back here."""
        pop, best, sigma = self.evaluate_async(rewards)
        return pop, best.result(), sigma

    def _population_size(self):
        raise NotImplementedError

    def _materialise(self, parents, parent_idx_host, sigma):
        """Build this rank's rows of the population described by (parents[K,P], parent_idx[N])."""
        n = len(parent_idx_host)
        shard = self._shard(n)
        if self.noise == "numpy" and shard.world != 1:
            raise RuntimeError("noise='numpy' reproduces the reference's single global stream: run it on one process")
        lo, hi = shard.first, shard.first + shard.n_local
        cached = getattr(self, "_idx_cache", None)     # the parent map is the same every generation: upload it once
        if cached is None or cached[2] is not parent_idx_host:
            local_idx = np.ascontiguousarray(parent_idx_host[lo:hi], dtype=np.int32)
            if cached is None or cached[0].shape != local_idx.shape or not np.array_equal(cached[0], local_idx):
                cached = (local_idx.copy(), torch.from_numpy(local_idx).to(self.dev.device), parent_idx_host)
            else:
                cached = (cached[0], cached[1], parent_idx_host)
            self._idx_cache = cached
        idx = cached[1]
        self._last = {"parents": parents, "idx_host": parent_idx_host, "sigma": sigma, "gen": self.gen, "shard": shard}
        if shard.n_local == 0:                      # more ranks than offspring: this rank idles through the rollout
            theta = self.dev.empty(0, self.P)
        elif self.noise == "philox":
            theta = self.dev.perturb(parents, sigma, self.seed, self.gen, lo, shard.n_local, parent_idx=idx)
        else:
            eps = np.zeros((n, self.P))
            for i in range(n):                     # the reference draws only for perturbed members, in order
                if parent_idx_host[i] >= 0:
                    eps[i] = np.random.normal(size=self.P)
            eps64 = torch.from_numpy(eps).to(self.dev.device)
            theta, store = self.dev.perturb_host_noise(parents, eps64, sigma, parent_idx=idx, want_eps_store=True)
            self._last["eps_store"] = store
            self._last["theta"] = theta
        pop = Population(theta, shard, self.network, self.agent_ids, self.gen)
        self.gen += 1
        return pop

    def _rows(self, ids_host):
        """Parameter rows of the CURRENT population for global ids (any rank can rebuild any row)."""
        last = self._last
        ids = np.asarray(ids_host, dtype=np.int32)
        if self.noise == "numpy":
            return self.dev.gather_rows(last["theta"], torch.from_numpy(ids).to(self.dev.device))
        sel = np.ascontiguousarray(last["idx_host"][ids])
        K = last["parents"].shape[0] if last["parents"].dim() == 2 else 1
        if sel.size and (sel.max() >= K or sel.min() < -K):
            raise ValueError("parent map out of range")
        both = torch.from_numpy(np.stack([sel, ids])).to(self.dev.device)          # one upload for both index rows
        return self.dev.perturb(last["parents"], last["sigma"], self.seed, last["gen"], 0, len(ids),
                                parent_idx=both[0], row_ids=both[1], idx_in_range=True)

    def _select_elites(self, rank, k, alias_state=None):
        """Elite ids, their rows and (simple_evolution) the aliasing flags without a host round trip: one
        ses_elite_select launch on the rank vector and the device copy of the current parent map."""
        last = self._last
        cached = getattr(self, "_map_dev", None)          # whole-population map on the device, uploaded once
        if cached is None or (cached[0] is not last["idx_host"] and not np.array_equal(cached[0], last["idx_host"])):
            cached = (last["idx_host"], torch.from_numpy(np.ascontiguousarray(last["idx_host"], dtype=np.int32)).to(self.dev.device))
            self._map_dev = cached
        ids, sel, alias = self.dev.elite_select(rank, k, cached[1], alias_state)
        self._elite_ids_dev = ids
        if self.noise == "numpy":
            rows = self.dev.gather_rows(last["theta"], ids)
        else:
            # the elite rows are not "the next population": this launch must not write the loop's tail time stamp
            # (ses_set_stamp), or eval_t could be read before the population-writing launch has stamped it
            keep = getattr(self.dev, "_stamp_keepalive", None)
            if keep is not None:
                self.dev.set_stamp(None)
            rows = self.dev.perturb(last["parents"], last["sigma"], self.seed, last["gen"], 0, k,
                                    parent_idx=sel, row_ids=ids, idx_in_range=True)   # map checked at upload
            if keep is not None:
                self.dev.set_stamp(keep)
        return rows, alias

    @property
    def elite_ids(self):
        """Global ids of the last elite selection (host copy on demand; the loop itself never reads them back)."""
        return None if self._elite_ids_dev is None else self._elite_ids_dev.cpu().numpy()

    def _fitness_tensor(self, rewards):
        if (isinstance(rewards, torch.Tensor) and rewards.dtype == torch.float32 and rewards.device == self.dev.device
                and rewards.is_contiguous() and rewards.numel() == self._population_size()):
            return rewards.view(-1)
        if isinstance(rewards, torch.Tensor):
            fit = rewards.to(device=self.dev.device, dtype=torch.float32).contiguous()
        else:
            fit = torch.as_tensor(np.asarray(rewards, dtype=np.float32)).to(self.dev.device)
        if fit.numel() != self._population_size():
            raise ValueError(f"expected {self._population_size()} rewards, got {fit.numel()}")
        return fit

    def _model_from(self, vec):
        import copy
        return copy.deepcopy(self.network).load_flat(vec.detach().cpu().numpy())

    # ---- roll-back support (ESLoop.run: a multi-GPU exchange that timed out is replayed from the last boundary) -------
    def snapshot(self, population):
        """Everything the next generations depend on, as device clones + host scalars, taken while `population` (the
        group evaluate() has just returned) is the current one.  restore() rebuilds that population bit for bit: the
        noise is a pure function of (seed, generation key, row, column)."""
        snap = {"curr_sigma": self.curr_sigma, "pop_gen": population.gen, "pop_sigma": self._last["sigma"]}
        snap.update(self._snapshot_state())
        return snap

    def restore(self, snap):
        if self.noise != "philox":
            raise RuntimeError("restore() needs counter-based noise (noise='philox')")
        self.curr_sigma = snap["curr_sigma"]
        self.gen = snap["pop_gen"]                 # _materialise keys the rows with it and bumps it again
        return self._restore_state(snap)


class simple_genetic(_DeviceStrategy):
    def __init__(self, init_sigma, sigma_decay, elite_num, offspring_num, noise="philox", seed=0):
        super().__init__(init_sigma, sigma_decay, offspring_num, noise, seed)
        self.elite_num = elite_num
        self.elite_models = None      # device float32[k, P], best first

    def _population_size(self):
        return self.elite_num * (self.offspring_num // self.elite_num)

    def _gen_offsprings(self, agent_ids, elite_models, elite_num, offspring_num, sigma):
        def build():
            per = offspring_num // elite_num
            idx = np.empty(elite_num * per, dtype=np.int32)
            for e in range(elite_num):
                idx[e * per] = -1 - e                 # the elite itself, verbatim
                idx[e * per + 1:(e + 1) * per] = e    # its children
            return idx
        return self._materialise(elite_models, self._const_map(("genetic", elite_num, offspring_num), build), sigma)

    def get_elite_model(self):
        return self._model_from(self.elite_models[0])

    def init_offspring(self, network, agent_ids):
        self._bind(network, agent_ids)
        self.elite_models = self.dev.zeros(self.elite_num, self.P)
        return self._gen_offsprings(agent_ids, self.elite_models, self.elite_num, self.offspring_num, self.curr_sigma)

    def evaluate_async(self, rewards):
        """evaluate() without the read-back: best_reward comes as a PendingReward (result() waits for it)."""
        fit = self._fitness_tensor(rewards)
        rank, _ = self.dev.rank_center(fit, want_weights=False, best=self._ring.arm())
        best = self._ring.push()
        self.elite_models, _ = self._select_elites(rank, self.elite_num)
        pop = self._gen_offsprings(self.agent_ids, self.elite_models, self.elite_num, self.offspring_num,
                                   self.curr_sigma)
        self.curr_sigma *= self.sigma_decay       # decays AFTER regeneration (offspring_strategies.py:117-124)
        return pop, best, self.curr_sigma

    def _snapshot_state(self):
        return {"elite_models": self.elite_models.clone()}

    def _restore_state(self, snap):
        self.elite_models = snap["elite_models"].clone()
        return self._gen_offsprings(self.agent_ids, self.elite_models, self.elite_num, self.offspring_num, snap["pop_sigma"])

    def get_wandb_cfg(self):
        return dict(init_sigma=self.init_sigma, sigma_decay=self.sigma_decay, elite_num=self.elite_num,
                    offspring_num=self.offspring_num)


class simple_evolution(_DeviceStrategy):
    def __init__(self, init_sigma, sigma_decay, elite_num, offspring_num, noise="philox", seed=0):
        super().__init__(init_sigma, sigma_decay, offspring_num, noise, seed)
        self.elite_num = elite_num
        self.mu_model = None          # device float32[P]
        self.elite0 = None            # device float32[P]  (elite_models[0])
        self._alias_state = None      # device int32[1]: population slots 0 and 1 are the same module object

    def _population_size(self):
        return self.offspring_num + 1

    def _gen_offsprings(self, agent_ids, elite_models, mu_model, sigma, offspring_num):
        # after evaluate() the elite IS the mean (the reference's in-place sum, offspring_strategies.py:234-248), and at
        # the start both are the zero network: one parent row then serves slots 0 and 1 -- no torch.stack (20 us of host
        # time and a copy kernel per generation, a tenth of a 96-offspring generation)
        same = elite_models is mu_model or elite_models.data_ptr() == mu_model.data_ptr()
        parents = mu_model.view(1, -1) if same else torch.stack([mu_model, elite_models]).contiguous()

        def build():
            idx = np.zeros(offspring_num + 1, dtype=np.int32)
            idx[0], idx[1] = -1, (-1 if same else -2)    # [mu, elite0, then N-1 children of mu]
            return idx
        return self._materialise(parents, self._const_map(("evolution", offspring_num, same), build), sigma)

    def get_elite_model(self):
        return self._model_from(self.elite0)

    def init_offspring(self, network, agent_ids):
        self._bind(network, agent_ids)
        self.mu_model = self.dev.zeros(self.P)
        self.elite0 = self.dev.zeros(self.P)
        self._alias_state = torch.ones(1, dtype=torch.int32, device=self.dev.device)
        return self._gen_offsprings(agent_ids, self.elite0, self.mu_model, self.curr_sigma, self.offspring_num)

    def evaluate_async(self, rewards):
        """evaluate() without the read-back: best_reward comes as a PendingReward (result() waits for it)."""
        fit = self._fitness_tensor(rewards)
        rank, _ = self.dev.rank_center(fit, want_weights=False, best=self._ring.arm())
        best = self._ring.push()
        # the reference sums the elites IN PLACE into elite[0]; an elite that is the same object as elite[0]
        # (slots 0 and 1 while they alias) doubles the running sum instead of adding its own value: the flags and
        # the aliasing state are kept on the device by ses_elite_select (include/ses.h)
        rows, alias = self._select_elites(rank, self.elite_num, self._alias_state)
        mean = self.dev.elite_mean(rows, alias)
        self.mu_model = mean
        self.elite0 = mean                         # elite[0] was overwritten with the mean (aliasing quirk)
        self.curr_sigma *= self.sigma_decay
        pop = self._gen_offsprings(self.agent_ids, self.elite0, self.mu_model, self.curr_sigma, self.offspring_num)
        return pop, best, self.curr_sigma

    def _snapshot_state(self):
        same = self.elite0 is self.mu_model or self.elite0.data_ptr() == self.mu_model.data_ptr()
        return {"mu": self.mu_model.clone(), "elite0": None if same else self.elite0.clone(),
                "alias": self._alias_state.clone()}

    def _restore_state(self, snap):
        self.mu_model = snap["mu"].clone()
        self.elite0 = self.mu_model if snap["elite0"] is None else snap["elite0"].clone()
        self._alias_state = snap["alias"].clone()
        return self._gen_offsprings(self.agent_ids, self.elite0, self.mu_model, snap["pop_sigma"], self.offspring_num)

    def get_wandb_cfg(self):
        return dict(init_sigma=self.init_sigma, elite_num=self.elite_num, offspring_num=self.offspring_num)


class openai_es(_DeviceStrategy):
    def __init__(self, init_sigma, sigma_decay, learning_rate, offspring_num, noise="philox", seed=0, fused=True):
        super().__init__(init_sigma, sigma_decay, offspring_num, noise, seed)
        self.learning_rate = learning_rate
        self.mu_model = None
        self.optimizer = None
        self.fused = bool(fused)      # philox noise: the whole fitness loop through ses_openai_generation (four launches)
        self._spare = None

    def _population_size(self):
        return self.offspring_num

    def _gen_offsprings(self, agent_ids, mu_model, sigma, offspring_num):
        def build():
            idx = np.zeros(offspring_num, dtype=np.int32)
            idx[0] = -1                           # member 0 is the unperturbed mu (epsilon = 0)
            return idx
        return self._materialise(mu_model.view(1, -1), self._const_map(("openai", offspring_num), build), sigma)

    def get_elite_model(self):
        return self._model_from(self.mu_model)

    def init_offspring(self, network, agent_ids):
        self._bind(network, agent_ids)
        self.mu_model = self.dev.zeros(self.P)
        self.optimizer = Adam(self.mu_model, self.learning_rate)
        return self._gen_offsprings(agent_ids, self.mu_model, self.curr_sigma, self.offspring_num)

    def evaluate_async(self, rewards):
        """evaluate() without the read-back: best_reward comes as a PendingReward (result() waits for it)."""
        fit = self._fitness_tensor(rewards)
        if self.noise == "philox" and self.fused:
            return self._evaluate_fused(fit)
        _, weights = self.dev.rank_center(fit, best=self._ring.arm())
        best = self._ring.push()
        a = self.optimizer.next_step_scale()
        opt = self.optimizer
        if self.noise == "philox":
            self.dev.es_update_philox(weights, self.seed, self._last["gen"], self.learning_rate, self.curr_sigma, a,
                                      self.mu_model, opt.m, opt.v, skip_row0=True)
        else:
            self.dev.es_update_stored(weights, self._last["eps_store"], self.learning_rate, self.curr_sigma, a,
                                      self.mu_model, opt.m, opt.v)
        self.curr_sigma *= self.sigma_decay
        pop = self._gen_offsprings(self.agent_ids, self.mu_model, self.curr_sigma, self.offspring_num)
        return pop, best, self.curr_sigma

    def _evaluate_fused(self, fit):
        """The same generation through ses_openai_generation: rank shaping, gradient, Adam and the next population
        in four launches; (mu, m, v) ping-pong between two buffer triples.  Bit-identical to the step-by-step path
        (tests/test_gpu_host_mirror.py::test_openai_fused_generation_equals_stepwise)."""
        opt = self.optimizer
        a = opt.next_step_scale()
        if self._spare is None:
            self._spare = tuple(torch.empty_like(self.mu_model) for _ in range(3))
        state_in, state_out = (self.mu_model, opt.m, opt.v), self._spare
        sigma = self.curr_sigma
        self.curr_sigma *= self.sigma_decay
        n = self.offspring_num
        shard = self._shard(n)
        comm = self._sharded_tail_comm(n, shard)
        theta = self.dev.openai_generation(fit, self.seed, self._last["gen"], self.learning_rate, sigma, a, state_in,
                                           state_out, self.curr_sigma, self.gen, shard.first, shard.n_local,
                                           best=self._ring.arm(), comm=comm, per_rank=shard.per_rank, world=shard.world)
        best = self._ring.push()
        self._spare = state_in
        self.mu_model, opt.m, opt.v = state_out
        opt.pi = self.mu_model
        self._last = {"parents": self.mu_model.view(1, -1), "idx_host": self._last["idx_host"], "sigma": self.curr_sigma,
                      "gen": self.gen, "shard": shard}
        pop = Population(theta, shard, self.network, self.agent_ids, self.gen)
        self.gen += 1
        return pop, best, self.curr_sigma

    def _sharded_tail_comm(self, n, shard):
        """The transport handle for ses_openai_generation_sharded, or None for the replicated tail: the shard form needs
        shards aligned to the gradient's 1024-row chunks and a library transport for the chunk partials (asked once per
        layout and transport state; the tuning knob "openai_sharded_tail" = 0 forces the replicated form, for A/B runs)."""
        owner = getattr(self.dev, "_comm_owner", None)
        if shard.world == 1 or owner is None:
            return None
        key = (n, shard.per_rank, shard.world, owner.comm_route())
        if getattr(self, "_sharded_key", None) != key:
            self._sharded_key = key
            self._sharded_ok = self.dev.openai_sharded_ok(owner, n, shard.per_rank, shard.world)
        return owner if self._sharded_ok else None

    def _snapshot_state(self):
        opt = self.optimizer
        return {"mu": self.mu_model.clone(), "m": opt.m.clone(), "v": opt.v.clone(), "t": opt.t}

    def _restore_state(self, snap):
        opt = self.optimizer
        self.mu_model, opt.m, opt.v, opt.t = snap["mu"].clone(), snap["m"].clone(), snap["v"].clone(), snap["t"]
        opt.pi = self.mu_model
        self._spare = None
        return self._gen_offsprings(self.agent_ids, self.mu_model, snap["pop_sigma"], self.offspring_num)

    def get_wandb_cfg(self):
        return dict(init_sigma=self.init_sigma, sigma_decay=self.sigma_decay, learning_rate=self.learning_rate,
                    offspring_num=self.offspring_num)
