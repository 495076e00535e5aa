"""``model_logging.Logger`` for the training loop of this repository (``WavenetTrainer.train`` calls ``logger.log(step,
loss)`` once per optimiser step).  It keeps the contract of the reference's class of the same name
(/root/reference/model_logging.py:12-58) -- constructor arguments, the three cadences, the ``log_loss`` / ``validate`` /
``generate`` hooks subclasses override, the printed lines -- and is written from that contract, not from its source.
``TensorboardLogger`` (:62-170, TensorFlow summaries) is host-side reporting outside this repository's scope.

Cadence (step numbers start at 1 in the trainer):
  every ``log_interval`` steps        -> ``log_loss(step)``: mean loss since the last report
  every ``validation_interval`` steps -> ``validate(step)``: ``trainer.validate()`` -> (loss, accuracy)
  every ``generate_interval`` steps   -> ``generate(step)``: ``generate_function(step)`` on a daemon thread, skipped while
                                         the previous one is still running (generation is slow next to a training step)
"""
import threading


class Logger:
    def __init__(self, log_interval=50, validation_interval=200, generate_interval=500, trainer=None, generate_function=None):
        self.trainer = trainer
        self.log_interval = log_interval
        self.validation_interval = validation_interval
        self.generate_interval = generate_interval
        self.generate_function = generate_function
        self.accumulated_loss = 0
        # a never-started placeholder, so that ``generate_thread.is_alive()`` is always answerable
        self.generate_thread = self._new_thread(()) if generate_function is not None else None

    def _new_thread(self, args):
        return threading.Thread(target=self.generate_function, args=args, daemon=True)

    def _due(self, step):
        """(hook, ...) whose interval divides ``step``, in reporting order."""
        table = ((self.log_interval, self._report_loss), (self.validation_interval, self.validate),
                 (self.generate_interval, self.generate))
        return [hook for every, hook in table if step % every == 0]

    def log(self, current_step, current_loss):
        self.accumulated_loss += current_loss
        for hook in self._due(current_step):
            hook(current_step)

    def _report_loss(self, current_step):
        self.log_loss(current_step)
        self.accumulated_loss = 0

    # ---- hooks (override in subclasses)
    def log_loss(self, current_step):
        print("loss at step %s: %s" % (current_step, self.accumulated_loss / self.log_interval))

    def validate(self, current_step):
        avg_loss, avg_accuracy = self.trainer.validate()
        print("validation loss: %s" % (avg_loss,))
        print("validation accuracy: %s%%" % (avg_accuracy * 100,))

    def generate(self, current_step):
        if self.generate_function is None:
            return
        if self.generate_thread.is_alive():
            print("Last generate is still running, skipping this one")
            return
        self.generate_thread = self._new_thread((current_step,))
        self.generate_thread.start()


class TensorboardLogger(Logger):
    def __init__(self, *a, **kw):
        raise NotImplementedError("TensorboardLogger writes TensorFlow summaries (model_logging.py:62-170): host-side reporting "
                                  "outside the MI355X hot-path scope; use Logger or subclass it")
