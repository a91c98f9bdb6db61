"""In-process stand-in for es_distributed/dist.py (Redis transport, dist.py:62-192).

Redis is out of scope for the hot path and not installed here, so MasterClient / WorkerClient
keep the reference's method names and task/result semantics over a thread-safe in-memory broker: the master
and the GPU worker(s) run as threads of one process.  Keys mirror dist.py:12-17; payloads are the same
namedtuples (not pickled -- nothing crosses a socket).  A maintainer who wants the real Redis path keeps
the reference's dist.py: the Task / Result wire types in es.py are unchanged.
"""
import threading
import time
from collections import deque

_brokers = {}
_brokers_lock = threading.Lock()


class _Broker:
    def __init__(self):
        self.cv = threading.Condition()
        self.exp = None
        self.task_id = -1
        self.task_data = None
        self.results = deque()
        self.archive = []


def _broker(redis_cfg):
    key = repr(sorted(redis_cfg.items())) if isinstance(redis_cfg, dict) else repr(redis_cfg)
    with _brokers_lock:
        if key not in _brokers:
            _brokers[key] = _Broker()
        return _brokers[key]


def reset_brokers():
    with _brokers_lock:
        _brokers.clear()


class MasterClient:
    """dist.py:62-98"""

    def __init__(self, master_redis_cfg):
        self.task_counter = 0
        self.b = _broker(master_redis_cfg)

    def declare_experiment(self, exp):
        with self.b.cv:
            self.b.exp = exp
            self.b.cv.notify_all()

    def declare_task(self, task_data):
        task_id = self.task_counter
        self.task_counter += 1
        with self.b.cv:
            self.b.task_id, self.b.task_data = task_id, task_data
            self.b.cv.notify_all()
        return task_id

    def pop_result(self):
        with self.b.cv:
            while not self.b.results:
                self.b.cv.wait()
            return self.b.results.popleft()

    def flush_results(self):
        with self.b.cv:
            n = len(self.b.results)
            self.b.results.clear()
            return n

    def add_to_novelty_archive(self, novelty_vector):
        with self.b.cv:
            self.b.archive.append(novelty_vector)

    def get_archive(self):
        with self.b.cv:
            return list(self.b.archive)


class WorkerClient:
    """dist.py:153-192.  Argument order as in the reference class: (relay_redis_cfg, master_redis_cfg)."""

    def __init__(self, relay_redis_cfg, master_redis_cfg):
        self.b = _broker(master_redis_cfg)

    def get_experiment(self):
        with self.b.cv:
            while self.b.exp is None:
                self.b.cv.wait()
            return self.b.exp

    def get_current_task(self):
        with self.b.cv:
            while self.b.task_data is None:
                self.b.cv.wait()
            return self.b.task_id, self.b.task_data

    def push_result(self, task_id, result):
        with self.b.cv:
            self.b.results.append((task_id, result))
            self.b.cv.notify_all()

    def get_archive(self):
        with self.b.cv:
            return list(self.b.archive)
