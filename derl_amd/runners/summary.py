"""Runner wrapper that switches summary recording on and off (derl/runners/summary.py)."""
from .env_runner import RunnerWrapper
from .. import summary


class PeriodicSummaries(RunnerWrapper):
  """Enables summary recording with the given period in env steps."""
  def __init__(self, runner, log_period):
    super().__init__(runner)
    self.log_period = log_period
    self.last_record_step = None

  @classmethod
  def make_with_nlogs(cls, runner, nlogs=1e5):
    if runner.nsteps is None:
      raise ValueError("runner.nsteps cannot be None")
    return cls(runner, int(runner.nsteps / nlogs))

  def run(self, obs=None):
    summary.start_recording()
    self.last_record_step = self.runner.step_count
    for interactions in self.runner.run(obs):
      yield interactions
      next_step = self.runner.step_count + 1
      should_record = next_step - self.last_record_step >= self.log_period
      summary.set_recording(should_record)
      if should_record:
        self.last_record_step = next_step
