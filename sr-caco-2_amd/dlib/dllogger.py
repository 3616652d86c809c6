"""Minimal message logger with the reference's call surface for this path
(``DLLogger.log(msg)`` / ``init_arb`` / ``flush``; the reference vendors NVIDIA's
DLLogger with an 'arbitrary message' extension, dlib/dllogger/logger.py:195-330): messages go
to stdout, ``log.txt`` (plain) and ``log.json`` (one ``DLLL {json}`` record per message) when a
directory is given, on the master process only."""
import datetime as dt
import json
import os

_state = {"txt": None, "json": None, "master": True, "stdout": True, "t0": dt.datetime.now()}


def init_arb(log_dir=None, is_master: bool = True, stdout: bool = True, reset: bool = True):
    for k in ("txt", "json"):
        if _state[k] is not None:
            _state[k].close()
            _state[k] = None
    _state.update(master=is_master, stdout=stdout, t0=dt.datetime.now())
    if log_dir is not None and is_master:
        os.makedirs(log_dir, exist_ok=True)
        _state["txt"] = open(os.path.join(log_dir, "log.txt"), "w" if reset else "a")
        _state["json"] = open(os.path.join(log_dir, "log.json"), "w" if reset else "a")


def log(message, verbosity: int = 1):
    if not _state["master"]:
        return
    now = dt.datetime.now()
    if _state["stdout"]:
        print(message, flush=True)
    if _state["txt"] is not None:
        _state["txt"].write(f"{message}\n")
        _state["txt"].flush()
    if _state["json"] is not None:
        rec = dict(timestamp=str(now.timestamp()), datetime=str(now), elapsedtime=str(now - _state["t0"]),
                   type="LOG", message=str(message))
        _state["json"].write("DLLL {}\n".format(json.dumps(rec)))
        _state["json"].flush()


def flush():
    for k in ("txt", "json"):
        if _state[k] is not None:
            _state[k].flush()
