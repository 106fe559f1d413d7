"""Switching summary recording on and off from the rollout loop -- the behaviour of
derl/runners/summary.py (``PeriodicSummaries``): recording is on for the first rollout, and after
each rollout it is on iff at least ``log_period`` env steps have passed since the rollout after
which it was last switched on (counting from the step after it)."""
from .. import summary
from .env_runner import RunnerWrapper


class PeriodicSummaries(RunnerWrapper):
  """Runner wrapper that enables summary recording every ``log_period`` env steps."""
  def __init__(self, runner, log_period):
    super().__init__(runner)
    self.log_period = log_period
    self.last_record_step = None

  @classmethod
  def make_with_nlogs(cls, runner, nlogs=1e5):
    """``nlogs`` evenly spaced recordings over the runner's ``nsteps``."""
    total = runner.nsteps
    if total is None:
      raise ValueError("runner.nsteps cannot be None")
    return cls(runner, int(total / nlogs))

  def _due(self):
    upcoming = self.runner.step_count + 1
    if upcoming - self.last_record_step < self.log_period:
      return False
    self.last_record_step = upcoming
    return True

  def run(self, obs=None):
    summary.start_recording()
    self.last_record_step = self.runner.step_count
    for interactions in self.runner.run(obs):
      yield interactions
      summary.set_recording(self._due())
