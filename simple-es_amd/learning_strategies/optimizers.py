"""Adam for openai_es -- device-resident mirror of the reference optimizer
(learning_strategies/optimizers.py:7-57: beta1 = 0.99, beta2 = 0.999, epsilon = 1e-8).

The moments m, v and the parameter vector live on the GPU as float32[P]; the update itself is fused
into the ES-gradient kernel (ses_es_update_*), which reproduces the reference's float32-moments /
float64-step arithmetic.  What stays on the host is the scalar step scale
    a = stepsize * sqrt(1 - beta2^t) / (1 - beta1^t)          (optimizers.py:43-47)
"""
import math


class Optimizer(object):
    def __init__(self, pi, epsilon=1e-08):
        self.pi = pi            # device float32[P] parameter vector (updated in place by the kernel)
        self.epsilon = epsilon
        self.t = 0

    def next_step_scale(self):
        raise NotImplementedError


class Adam(Optimizer):
    def __init__(self, pi, stepsize, beta1=0.99, beta2=0.999):
        super().__init__(pi)
        if (beta1, beta2) != (0.99, 0.999):
            raise ValueError("the fused update kernel is built for the reference's beta1=0.99, beta2=0.999")
        self.stepsize = stepsize
        self.beta1 = beta1
        self.beta2 = beta2
        self.m = pi.new_zeros(pi.shape)
        self.v = pi.new_zeros(pi.shape)

    def next_step_scale(self):
        """advance t and return `a` for this update"""
        self.t += 1
        return self.stepsize * math.sqrt(1 - self.beta2 ** self.t) / (1 - self.beta1 ** self.t)

    def state_dict(self):
        return {"t": self.t, "m": self.m.cpu(), "v": self.v.cpu()}
