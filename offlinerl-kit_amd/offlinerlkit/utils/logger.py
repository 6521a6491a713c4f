"""Key-value logger with the reference's surface (offlinerlkit/utils/logger.py:246-364): ``logkv``, ``logkv_mean``
(running mean per key per epoch), ``dumpkvs``, ``set_timestep``, ``log``, the ``record / checkpoint / model /
result`` directory layout and the ``policy_training_progress.csv`` format that the reference's plotting tools
read.  Sinks: aligned table on stdout / text file, CSV with a growing header, optional TensorBoard (skipped when
the package is absent).  Not on the GPU path: MFPolicyTrainer calls it once per epoch.
"""
from __future__ import annotations

import argparse
import datetime
import json
import os
import sys
from collections import defaultdict
from typing import Any, Dict, List, Optional, Sequence, Tuple, Union

DEBUG, INFO, WARN, ERROR, BACKUP = 10, 20, 30, 40, 60      # level constants of the reference module (logger.py:16-20)
DEFAULT_X_NAME = "timestep"
ROOT_DIR = "logs"


def _fmt(v) -> str:
    if hasattr(v, "__float__") and not isinstance(v, (int, bool)):
        return "%-8.3g" % float(v)
    return str(v)


class _TableSink:
    """Aligned ``| key | value |`` table, to a stream or a ``.txt`` file."""

    name = "stdout"

    def __init__(self, target) -> None:
        if isinstance(target, str):
            self.stream, self._own = open(target + ".txt", "at"), True
            self.name = os.path.splitext(os.path.basename(target))[0]
        else:
            self.stream, self._own = target, False

    def write_row(self, kvs: Dict) -> None:
        rows = [(str(k)[:40], _fmt(v)[:30]) for k, v in sorted(kvs.items(), key=lambda kv: str(kv[0]))]
        if not rows:
            return
        kw, vw = max(len(k) for k, _ in rows), max(len(v) for _, v in rows)
        bar = "-" * (kw + vw + 7)
        self.stream.write("\n".join([bar] + ["| %-*s | %-*s |" % (kw, k, vw, v) for k, v in rows] + [bar]) + "\n")
        self.stream.flush()

    def write_text(self, s: str) -> None:
        self.stream.write(s + "\n")
        self.stream.flush()

    def close(self) -> None:
        if self._own:
            self.stream.close()


class _CsvSink:
    """One line per dump; new keys widen the header and pad earlier rows (logger.py:144-198 behaviour)."""

    def __init__(self, path: str) -> None:
        self.path = path + ".csv"
        self.name = os.path.splitext(os.path.basename(self.path))[0]
        self.keys: List[str] = []
        if os.path.exists(self.path):
            with open(self.path) as f:
                head = f.readline().rstrip("\n")
            self.keys = head.split(",") if head else []

    def write_row(self, kvs: Dict) -> None:
        new = sorted(k for k in kvs if k not in self.keys)
        if new:
            old_lines = []
            if os.path.exists(self.path):
                with open(self.path) as f:
                    old_lines = f.read().splitlines()[1:]
            self.keys += new
            with open(self.path, "w") as f:
                f.write(",".join(self.keys) + "\n")
                for ln in old_lines:
                    f.write(ln + "," * len(new) + "\n")
        with open(self.path, "a") as f:
            f.write(",".join("" if kvs.get(k) is None else str(kvs.get(k)) for k in self.keys) + "\n")

    def close(self) -> None:
        pass


class _TbSink:
    def __init__(self, path: str) -> None:
        from torch.utils.tensorboard import SummaryWriter  # raises when tensorboard is not installed
        self.name = os.path.basename(path)
        self.writer = SummaryWriter(path)
        self.step = 0

    def write_row(self, kvs: Dict) -> None:
        for k, v in kvs.items():
            if k != DEFAULT_X_NAME:
                self.writer.add_scalar(k, v, self.step)
        self.writer.flush()

    def close(self) -> None:
        self.writer.close()


class Logger:
    def __init__(self, dir: str, ouput_config: Dict) -> None:
        self._dir = dir
        self._record_dir = os.path.join(dir, "record")
        self._checkpoint_dir = os.path.join(dir, "checkpoint")
        self._model_dir = os.path.join(dir, "model")
        self._result_dir = os.path.join(dir, "result")
        for d in (self._record_dir, self._checkpoint_dir, self._model_dir, self._result_dir):
            os.makedirs(d, exist_ok=True)
        self._sinks: list = []
        for file_name, kind in ouput_config.items():
            path = os.path.join(self._record_dir, file_name)
            try:
                if kind == "stdout":
                    self._sinks.append(_TableSink(path))
                elif kind == "csv":
                    self._sinks.append(_CsvSink(path))
                elif kind == "tensorboard":
                    self._sinks.append(_TbSink(path))
            except Exception as e:  # tensorboard is optional in this environment
                sys.stderr.write(f"[Logger] skipping {kind} sink {file_name}: {e}\n")
        self._sinks.append(_TableSink(sys.stdout))
        self._name2val: Dict[Any, Any] = defaultdict(float)
        self._name2cnt: Dict[Any, int] = defaultdict(int)
        self._timestep = 0
        self._level = INFO

    def log_hyperparameters(self, hyper_param: Dict) -> None:
        with open(os.path.join(self._record_dir, "hyper_param.json"), "w") as f:
            json.dump({k: (v if isinstance(v, (int, float, str, bool, list, type(None))) else str(v)) for k, v in hyper_param.items()}, f, indent=4)

    def logkv(self, key: Any, val: Any) -> None:
        self._name2val[key] = val

    def logkv_mean(self, key: Any, val) -> None:
        old, cnt = self._name2val[key], self._name2cnt[key]
        self._name2val[key] = old * cnt / (cnt + 1) + val / (cnt + 1)
        self._name2cnt[key] = cnt + 1

    def dumpkvs(self, exclude: Optional[Union[str, Tuple[str, ...]]] = None) -> None:
        self.logkv(DEFAULT_X_NAME, self._timestep)
        for s in self._sinks:
            if exclude is not None and s.name in exclude:
                continue
            s.write_row(self._name2val)
        self._name2val.clear()
        self._name2cnt.clear()

    def log(self, s: str, level=INFO) -> None:
        # (like the reference, `level` is accepted and stored by set_level but does not filter: logger.py:311-314, 322-323)
        for sink in self._sinks:
            if isinstance(sink, _TableSink):
                sink.write_text(s)

    def set_timestep(self, timestep: int) -> None:
        self._timestep = timestep
        for s in self._sinks:
            if isinstance(s, _TbSink):
                s.step = timestep

    def set_level(self, level) -> None:
        self._level = level

    record_dir = property(lambda self: self._record_dir)
    checkpoint_dir = property(lambda self: self._checkpoint_dir)
    model_dir = property(lambda self: self._model_dir)
    result_dir = property(lambda self: self._result_dir)

    def close(self) -> None:
        for s in self._sinks:
            s.close()


def make_log_dirs(task_name: str, algo_name: str, seed: Union[int, str], args: Dict, part: Optional[str] = None,
                  record_params: Optional[Sequence[str]] = None) -> str:
    """logs/<task>/<algo>[&param=value...]/[part/]timestamp_<yy-mmdd-HHMMSS>&<seed> (logger.py:346-364)."""
    for name in record_params or ():
        algo_name += f"&{name}={args[name]}"
    stamp = datetime.datetime.now().strftime("%y-%m%d-%H%M%S")
    parts = [ROOT_DIR, task_name, algo_name] + ([part] if part is not None else []) + [f"timestamp_{stamp}&{seed}"]
    path = os.path.join(*parts)
    os.makedirs(path)
    return path


def load_args(load_path: str) -> argparse.Namespace:
    """hyper_param.json (written by ``log_hyperparameters``) back as the Namespace a launch script's ``get_args()`` returns
    (logger.py:367-371)."""
    with open(load_path, "r") as f:
        return argparse.Namespace(**json.load(f))
