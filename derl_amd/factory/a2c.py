"""A2C factory: the hyper-parameters and wiring of derl/factory/a2c.py:15-80 (atari preset only,
GAE with lambda 1 by default, RMSprop, `lr` annealed linearly to zero) building this package's
device-resident objects."""
from ..alg.a2c import A2C
from ..alg.common import Trainer
from ..anneal import LinearAnneal
from ..models import make_model
from ..optim import RMSprop
from ..policies import ActorCriticPolicy, sampling_seed
from ..runners import env_runner, onpolicy, summary as runner_summary, trajectory_transforms
from .factory import Factory

_ATARI = {  # derl/factory/a2c.py:22-37
    "nenvs": 8, "num-train-steps": 10e6, "num-runner-steps": 5, "gamma": 0.99, "lambda_": 1.,
    "normalize-gae": dict(action="store_true"), "lr": 7e-4, "optimizer-alpha": 0.99,
    "optimizer-epsilon": 1e-5, "value-loss-coef": 0.5, "entropy-coef": 0.01, "max-grad-norm": 0.5,
}
_SKIP_CHECK = ("nenvs",)


class A2CFactory(Factory):
  """Advantage Actor-Critic factory."""
  def __init__(self, *, ignore_unused=_SKIP_CHECK, **kwargs):
    super().__init__(ignore_unused=ignore_unused, **kwargs)

  @staticmethod
  def get_parser_defaults(args_type="atari"):
    return dict(_ATARI) if args_type == "atari" else None

  @classmethod
  def from_default_kwargs(cls, args_type="atari", ignore_unused=_SKIP_CHECK, **kwargs):
    return super().from_default_kwargs(args_type, ignore_unused, **kwargs)

  @classmethod
  def from_args(cls, args_type="atari", ignore_unused=_SKIP_CHECK, args=None):
    return super().from_args(args_type, ignore_unused, args)

  def make_runner(self, env, nlogs=1e5, **kwargs):
    with self.override_context(**kwargs):
      model = self.get_arg("model") if self.has_arg("model") else make_model(
          env.observation_space, env.action_space, 1)
      policy = ActorCriticPolicy(model, seed=sampling_seed(env))
      steps = env_runner.EnvRunner(env, policy, self.get_arg("num_runner_steps"),
                                   nsteps=self.get_arg("num_train_steps"))
      logged = runner_summary.PeriodicSummaries.make_with_nlogs(steps, nlogs)
      gae = trajectory_transforms.GAE(policy, normalize=self.get_arg_default("normalize_gae", False),
                                      **self.get_arg_dict("gamma", "lambda_"))
      batched = hasattr(env.unwrapped, "nenvs")
      transforms = [gae, trajectory_transforms.MergeTimeBatch()] if batched else [gae]
      return onpolicy.TransformInteractions(logged, transforms)

  def make_trainer(self, runner, **kwargs):
    with self.override_context(**kwargs):
      schedule = LinearAnneal(self.get_arg("lr"), self.get_arg("num_train_steps"), 0., name="lr")
      # 1e-55 is the reference's fallback epsilon (factory/a2c.py:72), kept as is
      rmsprop = RMSprop(runner.policy.model, schedule.get_tensor(),
                        alpha=self.get_arg_default("optimizer_alpha", 0.99),
                        eps=self.get_arg_default("optimizer_epsilon", 1e-55))
      return Trainer(rmsprop, anneals=[schedule], max_grad_norm=self.get_arg_default("max_grad_norm"))

  def make_alg(self, runner, trainer, **kwargs):
    with self.override_context(**kwargs):
      return A2C(runner, trainer, **self.get_arg_dict("value_loss_coef", "entropy_coef"))
