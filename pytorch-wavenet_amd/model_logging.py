"""Drop-in for the reference's ``model_logging.Logger`` (/root/reference/model_logging.py:12-58): loss / validation /
background-generation cadence of the training loop.  ``TensorboardLogger`` (:62-170) is a TensorFlow summary writer --
host-side reporting outside this repository's scope -- and is not provided."""
import threading


class Logger:
    def __init__(self, log_interval=50, validation_interval=200, generate_interval=500, trainer=None, generate_function=None):
        self.trainer = trainer
        self.log_interval = log_interval
        self.validation_interval = validation_interval
        self.generate_interval = generate_interval
        self.accumulated_loss = 0
        self.generate_function = generate_function
        if self.generate_function is not None:
            self.generate_thread = threading.Thread(target=self.generate_function)
            self.generate_thread.daemon = True  # upstream sets .daemon on the function object (:26), a no-op

    def log(self, current_step, current_loss):  # :29-37
        self.accumulated_loss += current_loss
        if current_step % self.log_interval == 0:
            self.log_loss(current_step)
            self.accumulated_loss = 0
        if current_step % self.validation_interval == 0:
            self.validate(current_step)
        if current_step % self.generate_interval == 0:
            self.generate(current_step)

    def log_loss(self, current_step):
        avg_loss = self.accumulated_loss / self.log_interval
        print("loss at step " + str(current_step) + ": " + str(avg_loss))

    def validate(self, current_step):
        avg_loss, avg_accuracy = self.trainer.validate()
        print("validation loss: " + str(avg_loss))
        print("validation accuracy: " + str(avg_accuracy * 100) + "%")

    def generate(self, current_step):  # :48-58: one background generation at a time
        if self.generate_function is None:
            return
        if self.generate_thread.is_alive():
            print("Last generate is still running, skipping this one")
        else:
            self.generate_thread = threading.Thread(target=self.generate_function, args=[current_step])
            self.generate_thread.daemon = True
            self.generate_thread.start()


class TensorboardLogger(Logger):
    def __init__(self, *a, **kw):
        raise NotImplementedError("TensorboardLogger writes TensorFlow summaries (model_logging.py:62-170): host-side reporting "
                                  "outside the MI355X hot-path scope; use Logger or subclass it")
