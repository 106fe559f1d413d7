"""PPO factory: the hyper-parameters and wiring of derl/factory/ppo.py:12-91 (atari and mujoco
presets, `lr` annealed linearly to zero over `num-train-steps`, Adam with `optimizer-epsilon`,
global-norm clipping) building this package's device-resident objects."""
from ..alg.common import Trainer
from ..alg.ppo import PPO
from ..anneal import LinearAnneal
from ..models import make_model
from ..optim import Adam
from ..policies import ActorCriticPolicy, sampling_seed
from ..runners.onpolicy import make_ppo_runner
from .factory import Factory

# flag -> (atari, mujoco); derl/factory/ppo.py:19-49
_PRESETS = (
    ("num-train-steps", 10e6, 1e6),
    ("nenvs", 8, dict(type=int, default=None)),
    ("num-runner-steps", 128, 2048),
    ("gamma", 0.99, 0.99),
    ("lambda_", 0.95, 0.95),
    ("num-epochs", 3, 10),
    ("num-minibatches", 4, 32),
    ("cliprange", 0.1, 0.2),
    ("value-loss-coef", 0.25, 0.25),
    ("entropy-coef", 0.01, 0.),
    ("max-grad-norm", 0.5, 0.5),
    ("lr", 2.5e-4, 3e-4),
    ("optimizer-epsilon", 1e-5, 1e-5),
)
_COLUMN = {"atari": 1, "mujoco": 2}
_SKIP_CHECK = ("nenvs",)  # consumed by env construction, not by the factory


class PPOFactory(Factory):
  """Proximal Policy Optimization factory."""
  def __init__(self, *, ignore_unused=_SKIP_CHECK, **kwargs):
    super().__init__(ignore_unused=ignore_unused, **kwargs)

  @staticmethod
  def get_parser_defaults(args_type="atari"):
    column = _COLUMN.get(args_type)
    return None if column is None else {row[0]: row[column] for row in _PRESETS}

  @classmethod
  def from_default_kwargs(cls, args_type="atari", ignore_unused=_SKIP_CHECK, **kwargs):
    return super().from_default_kwargs(args_type, ignore_unused, **kwargs)

  @classmethod
  def from_args(cls, args_type="atari", ignore_unused=_SKIP_CHECK, args=None):
    return super().from_args(args_type, ignore_unused, args)

  def _policy(self, env):
    model = self.get_arg("model") if self.has_arg("model") else make_model(
        env.observation_space, env.action_space, 1)
    return ActorCriticPolicy(model, seed=sampling_seed(env))

  def make_runner(self, env, nlogs=1e5, **kwargs):
    with self.override_context(**kwargs):
      horizon, total = self.get_arg_list("num_runner_steps", "num_train_steps")
      wrap = self.get_arg_dict("gamma", "lambda_", "num_epochs", "num_minibatches")
      return make_ppo_runner(env, self._policy(env), horizon, total, nlogs=nlogs, **wrap)

  def make_trainer(self, runner, **kwargs):
    with self.override_context(**kwargs):
      schedule = LinearAnneal(self.get_arg("lr"), self.get_arg("num_train_steps"), name="lr")
      adam = dict(lr=schedule.get_tensor())
      if self.has_arg("optimizer_epsilon"):
        adam["eps"] = self.get_arg("optimizer_epsilon")
      return Trainer(Adam(runner.policy.model, **adam), anneals=[schedule],
                     max_grad_norm=self.get_arg("max_grad_norm"))

  def make_alg(self, runner, trainer, **kwargs):
    with self.override_context(**kwargs):
      return PPO(runner, trainer, **self.get_arg_dict("value_loss_coef", "entropy_coef", "cliprange"))
