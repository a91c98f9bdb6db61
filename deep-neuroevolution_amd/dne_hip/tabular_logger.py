"""Minimal tabular logger with the reference's call surface (es_distributed/tabular_logger.py:60-74,131-153):
start / log / record_tabular / dump_tabular.  stdout table + log.txt; TensorBoard events are out of scope."""
import os
import sys
from collections import OrderedDict

_state = {"dir": None, "file": None, "row": OrderedDict()}


def start(log_dir):
    os.makedirs(log_dir, exist_ok=True)
    _state["dir"] = log_dir
    _state["file"] = open(os.path.join(log_dir, "log.txt"), "a")


def stop():
    if _state["file"]:
        _state["file"].close()
    _state["file"] = None


def log(*args):
    line = " ".join(str(a) for a in args)
    print(line, file=sys.stdout)
    if _state["file"]:
        _state["file"].write(line + "\n")
        _state["file"].flush()


def record_tabular(key, val):
    _state["row"][key] = val


def dump_tabular():
    row = _state["row"]
    if not row:
        return
    kw = max(len(k) for k in row)
    lines = ["-" * (kw + 22)]
    for k, v in row.items():
        vs = "%-8.6g" % v if isinstance(v, float) or hasattr(v, "dtype") else str(v)
        lines.append("| %s | %-15s |" % (k.ljust(kw), vs))
    lines.append("-" * (kw + 22))
    log("\n".join(lines))
    row.clear()
