"""numpy restatement of the reference's offspring strategies and Adam.

TEST INFRASTRUCTURE (oracle).  Each individual is ONE flat float32 vector in
`parameters()` order (networks/neural_network.py:46-56) instead of a
torch module; every arithmetic statement keeps the reference's numpy dtypes
and in-place semantics, so under the numpy installed here (2.2, NEP 50
promotion) the results are bit-identical to the reference -- this is pinned by
tests/test_oracle_golden.py against fixtures produced by importing the
reference (tests/golden/make_golden.py).

Restates (file:line into /root/reference/learning_strategies):
  simple_genetic     evolution/offspring_strategies.py:11-134
  simple_evolution   evolution/offspring_strategies.py:137-267
  openai_es          evolution/offspring_strategies.py:270-434
  Adam               optimizers.py:7-57

Object identity matters in the reference: a population slot is a
reference to a module, several slots can be the SAME module, and
`simple_evolution.evaluate` sums elites in place into elite[0].  Individuals
are therefore kept as numpy array objects and combined with the same in-place
operators, which reproduces the aliasing without special cases.

`noise` hook: callable(P, i) -> float64[P] standard normals for the i-th draw;
default draws from the global legacy numpy generator exactly like the
reference (`np.random.normal(size=shape)` per tensor == one flat stream).
"""
import numpy as np


def _np_noise(P, _i):
    return np.random.normal(size=P)


def rank_desc(rewards, stable=False):
    """np.flip(np.argsort(rewards)) (offspring_strategies.py:112,234,380).

    stable=True is the tie rule the device kernels implement
    (reward descending, then index descending); for tie-free input both agree."""
    r = np.array(rewards)
    return np.flip(np.argsort(r, kind="stable" if stable else None))


def centered_ranks(rewards, stable=False):
    """offspring_strategies.py:380-398 -> float64[n] shaped rewards."""
    order = rank_desc(rewards, stable)
    n = len(rewards)
    reward_array = np.zeros(n)
    for idx in reversed(range(n)):
        reward_array[order[idx]] = ((n - 1 - idx) / (n - 1)) - 0.5
    r_std = reward_array.std()
    return (reward_array - reward_array.mean()) / r_std


class AdamNP:
    """optimizers.py:30-57 on one flat vector (beta1 = 0.99 as in the reference)."""

    def __init__(self, theta, stepsize, beta1=0.99, beta2=0.999, epsilon=1e-08):
        self.theta = theta              # float32[P], updated in place
        self.stepsize = stepsize
        self.beta1 = beta1
        self.beta2 = beta2
        self.epsilon = epsilon
        self.t = 0
        self.m = np.zeros_like(theta)
        self.v = np.zeros_like(theta)

    def step_scale(self):
        return self.stepsize * np.sqrt(1 - self.beta2 ** self.t) / (1 - self.beta1 ** self.t)

    def update(self, grad):
        self.t += 1
        a = self.step_scale()
        self.m = self.beta1 * self.m + (1 - self.beta1) * grad
        self.v = self.beta2 * self.v + (1 - self.beta2) * (grad * grad)
        step = -a * self.m / (np.sqrt(self.v) + self.epsilon)
        self.theta += step
        return step


class OpenAIESNP:
    def __init__(self, P, init_sigma, sigma_decay, learning_rate, offspring_num, noise=_np_noise, stable_rank=False):
        self.P = P
        self.offspring_num = offspring_num
        self.sigma_decay = sigma_decay
        self.learning_rate = learning_rate
        self.curr_sigma = init_sigma
        self.noise = noise
        self.stable_rank = stable_rank
        self.mu = np.zeros(P, dtype=np.float32)          # zero_init, loop.py:31
        self.optimizer = AdamNP(self.mu, learning_rate)
        self.epsilons = []
        self.draws = 0
        self.population = self._gen()

    def _gen(self):
        self.epsilons = [self.mu.copy()]                 # :303-308 "zero" net keeps mu's values
        pop = [self.mu.copy()]                           # :310 member 0 = mu
        for _ in range(self.offspring_num - 1):
            epsilon = self.noise(self.P, self.draws)
            self.draws += 1
            eps_param = self.mu.copy()                   # :316 deepcopy(zero_net_param_list)
            perturb = self.mu.copy()
            eps_param += epsilon                         # :321
            perturb += epsilon * self.curr_sigma         # :322
            pop.append(perturb)
            self.epsilons.append(eps_param)
        return pop

    def theta(self):
        return np.stack(self.population)

    def evaluate(self, rewards):
        best_reward = max(rewards)
        reward_array = centered_ranks(rewards, self.stable_rank)
        grad = np.zeros(self.P, dtype=np.float32)        # :401-404
        update_factor = self.learning_rate / (len(self.epsilons) * self.curr_sigma)
        update_factor *= -1.0
        for offs_idx, offs in enumerate(self.epsilons):
            grad += offs * reward_array[offs_idx]        # :412
        grad *= update_factor                            # :414
        self.last_grad = grad.copy()
        self.last_weights = reward_array
        self.optimizer.update(grad)                      # :416
        self.curr_sigma *= self.sigma_decay              # :418
        self.population = self._gen()
        return best_reward, self.curr_sigma


class SimpleEvolutionNP:
    def __init__(self, P, init_sigma, sigma_decay, elite_num, offspring_num, noise=_np_noise, stable_rank=False):
        self.P = P
        self.elite_num = elite_num
        self.offspring_num = offspring_num
        self.sigma_decay = sigma_decay
        self.curr_sigma = init_sigma
        self.noise = noise
        self.stable_rank = stable_rank
        self.draws = 0
        net = np.zeros(P, dtype=np.float32)
        self.elite_models = [net for _ in range(elite_num)]   # :201 same object k times
        self.mu_model = self.elite_models[0]
        self.population = self._gen()

    def _gen(self):
        pop = [self.mu_model, self.elite_models[0]]      # :166-167 references, no copy
        for _ in range(self.offspring_num - 1):
            child = self.mu_model.copy()
            epsilon = self.noise(self.P, self.draws) * self.curr_sigma   # normal(0, sigma) = sigma * z
            self.draws += 1
            child += epsilon                             # :174
            pop.append(child)
        return pop

    def theta(self):
        return np.stack(self.population)

    def evaluate(self, rewards):
        elite_ids = rank_desc(rewards, self.stable_rank)[: self.elite_num]
        best_reward = max(rewards)
        self.elite_ids = np.array(elite_ids)
        self.elite_models = [self.population[i] for i in elite_ids]
        new_mu = self.elite_models[0]                    # :241 views of elite[0]: in-place sum
        for elite in self.elite_models[1:]:
            new_mu += elite                              # :245 (aliasing doubles when elite is elite[0])
        new_mu /= self.elite_num                         # :248
        # :250 apply_param copies the values into mu_model (which keeps its own identity)
        if self.mu_model is not new_mu:
            self.mu_model[...] = new_mu
        self.curr_sigma *= self.sigma_decay              # :251
        self.population = self._gen()
        return best_reward, self.curr_sigma


class SimpleGeneticNP:
    def __init__(self, P, init_sigma, sigma_decay, elite_num, offspring_num, noise=_np_noise, stable_rank=False):
        self.P = P
        self.elite_num = elite_num
        self.offspring_num = offspring_num
        self.sigma_decay = sigma_decay
        self.curr_sigma = init_sigma
        self.noise = noise
        self.stable_rank = stable_rank
        self.draws = 0
        net = np.zeros(P, dtype=np.float32)
        self.elite_models = [net for _ in range(elite_num)]   # :84
        self.population = self._gen()

    def _gen(self):
        pop = []
        for p in self.elite_models:
            pop.append(p)                                # :51 the elite itself
            for _ in range((self.offspring_num // self.elite_num) - 1):
                child = p.copy()
                child += self.noise(self.P, self.draws) * self.curr_sigma   # :57-58
                self.draws += 1
                pop.append(child)
        return pop

    def theta(self):
        return np.stack(self.population)

    def evaluate(self, rewards):
        elite_ids = rank_desc(rewards, self.stable_rank)[: self.elite_num]
        best_reward = max(rewards)
        self.elite_ids = np.array(elite_ids)
        self.elite_models = [self.population[i] for i in elite_ids]
        self.population = self._gen()                    # :117 uses the un-decayed sigma
        self.curr_sigma *= self.sigma_decay              # :124 decay AFTER regeneration
        return best_reward, self.curr_sigma
