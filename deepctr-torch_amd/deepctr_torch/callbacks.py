"""TensorFlow-free callbacks.  The reference re-exports tf.keras' ``EarlyStopping`` / ``History`` and
subclasses its ``ModelCheckpoint`` (deepctr_torch/callbacks.py:1-73); TensorFlow is not a dependency
here, so the small part of the Keras callback protocol that ``BaseModel.fit`` drives
(basemodel.py:220-227,233,303-307) is implemented directly.  Out of the hot path."""
import numpy as np
import torch


class Callback(object):
    def __init__(self):
        self.model = None
        self.params = None

    def set_params(self, params):
        self.params = params

    def set_model(self, model):
        self.model = model

    def on_train_begin(self, logs=None):
        pass

    def on_train_end(self, logs=None):
        pass

    def on_epoch_begin(self, epoch, logs=None):
        pass

    def on_epoch_end(self, epoch, logs=None):
        pass

    def on_batch_begin(self, batch, logs=None):
        pass

    def on_batch_end(self, batch, logs=None):
        pass


class CallbackList(object):
    def __init__(self, callbacks=None):
        self.callbacks = list(callbacks or [])
        self.model = None

    def append(self, callback):
        self.callbacks.append(callback)

    def set_model(self, model):
        self.model = model
        for cb in self.callbacks:
            cb.set_model(model)

    def _fan_out(self, hook, *args):
        for cb in self.callbacks:
            getattr(cb, hook)(*args)

    def on_train_begin(self, logs=None):
        self._fan_out("on_train_begin", logs)

    def on_train_end(self, logs=None):
        self._fan_out("on_train_end", logs)

    def on_epoch_begin(self, epoch, logs=None):
        self._fan_out("on_epoch_begin", epoch, logs)

    def on_epoch_end(self, epoch, logs=None):
        self._fan_out("on_epoch_end", epoch, logs)


class History(Callback):
    """Per-epoch log accumulator; ``model.history.history[name]`` is a list with one value per epoch."""

    def on_train_begin(self, logs=None):
        self.epoch = []
        self.history = {}

    def on_epoch_end(self, epoch, logs=None):
        self.epoch.append(epoch)
        for k, v in (logs or {}).items():
            self.history.setdefault(k, []).append(v)


def _direction(mode, monitor):
    if mode == 'min':
        return np.less
    if mode == 'max':
        return np.greater
    return np.greater if ('acc' in monitor or monitor.startswith('fmeasure') or 'auc' in monitor) else np.less


class EarlyStopping(Callback):
    """Stop when ``monitor`` has not improved by ``min_delta`` for ``patience`` epochs (Keras semantics)."""

    def __init__(self, monitor='val_loss', min_delta=0, patience=0, verbose=0, mode='auto', baseline=None,
                 restore_best_weights=False):
        super(EarlyStopping, self).__init__()
        self.monitor, self.patience, self.verbose, self.baseline = monitor, patience, verbose, baseline
        self.restore_best_weights = restore_best_weights
        self.monitor_op = _direction(mode, monitor)
        self.min_delta = abs(min_delta) * (1 if self.monitor_op == np.greater else -1)
        self.wait, self.stopped_epoch, self.best_weights = 0, 0, None

    def on_train_begin(self, logs=None):
        self.wait, self.stopped_epoch = 0, 0
        self.best = self.baseline if self.baseline is not None else (
            np.inf if self.monitor_op == np.less else -np.inf)

    def on_epoch_end(self, epoch, logs=None):
        current = (logs or {}).get(self.monitor)
        if current is None:
            print('Early stopping conditioned on metric `%s` which is not available. Available metrics are: %s'
                  % (self.monitor, ','.join(list((logs or {}).keys()))))
            return
        if self.monitor_op(current - self.min_delta, self.best):
            self.best, self.wait = current, 0
            if self.restore_best_weights:
                self.best_weights = {k: v.detach().clone() for k, v in self.model.state_dict().items()}
        else:
            self.wait += 1
            if self.wait >= self.patience:
                self.stopped_epoch = epoch
                self.model.stop_training = True
                if self.restore_best_weights and self.best_weights is not None:
                    self.model.load_state_dict(self.best_weights)

    def on_train_end(self, logs=None):
        if self.stopped_epoch > 0 and self.verbose > 0:
            print('Epoch %05d: early stopping' % (self.stopped_epoch + 1))


class ModelCheckpoint(Callback):
    """Save the model (``torch.save``) after every ``period`` epochs, optionally only when ``monitor``
    improved -- constructor arguments of the reference's checkpoint (callbacks.py:9-73)."""

    def __init__(self, filepath, monitor='val_loss', verbose=0, save_best_only=False, save_weights_only=False,
                 mode='auto', period=1):
        super(ModelCheckpoint, self).__init__()
        self.filepath, self.monitor, self.verbose = filepath, monitor, verbose
        self.save_best_only, self.save_weights_only, self.period = save_best_only, save_weights_only, period
        self.epochs_since_last_save = 0
        self.monitor_op = _direction(mode, monitor)
        self.best = np.inf if self.monitor_op == np.less else -np.inf

    def _save(self, filepath):
        torch.save(self.model.state_dict() if self.save_weights_only else self.model, filepath)

    def on_epoch_end(self, epoch, logs=None):
        logs = logs or {}
        self.epochs_since_last_save += 1
        if self.epochs_since_last_save < self.period:
            return
        self.epochs_since_last_save = 0
        filepath = self.filepath.format(epoch=epoch + 1, **logs)
        if not self.save_best_only:
            if self.verbose > 0:
                print('Epoch %05d: saving model to %s' % (epoch + 1, filepath))
            self._save(filepath)
            return
        current = logs.get(self.monitor)
        if current is None:
            print('Can save best model only with %s available, skipping.' % self.monitor)
        elif self.monitor_op(current, self.best):
            if self.verbose > 0:
                print('Epoch %05d: %s improved from %0.5f to %0.5f, saving model to %s'
                      % (epoch + 1, self.monitor, self.best, current, filepath))
            self.best = current
            self._save(filepath)
        elif self.verbose > 0:
            print('Epoch %05d: %s did not improve from %0.5f' % (epoch + 1, self.monitor, self.best))
