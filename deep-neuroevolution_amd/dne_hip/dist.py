"""Transport of es_distributed/dist.py (dist.py:62-192) for the HIP drivers: same classes (MasterClient, RelayClient,
WorkerClient), same Redis keys, same pickled payloads, so HIP workers can serve a reference master and a HIP master
can drive reference CPU workers.

Two carriers behind the same methods:
  * Redis -- the reference's own deployment.  The `redis` package is used when importable; otherwise the small RESP
    client in resp.py speaks to the server directly (this image has neither the package nor a server).
  * an in-process broker -- master and GPU worker(s) as threads of one process, no server, nothing pickled.
    Only on request: redis_cfg['transport'] = 'inprocess' (or DNE_TRANSPORT=inprocess in the environment).
A configuration that names a server (host / unix_socket_path) means Redis and behaves like the reference: 300 connection
attempts, then the error (dist.py:27-43) -- a worker that cannot reach its master must not quietly wait on a private broker.
transport = 'auto' (one attempt, then the in-process broker, with a warning) exists for single-process experiments.
"""
import logging
import os
import pickle
import threading
import time
from collections import deque

logger = logging.getLogger(__name__)

# dist.py:12-17
EXP_KEY = 'es:exp'
TASK_ID_KEY = 'es:task_id'
TASK_DATA_KEY = 'es:task_data'
TASK_CHANNEL = 'es:task_channel'
RESULTS_KEY = 'es:results'
ARCHIVE_KEY = 'es:archive'


def serialize(x):      # dist.py:19-20
    return pickle.dumps(x, protocol=-1)


def deserialize(x):    # dist.py:23-24
    return pickle.loads(x)


# ---------------------------------------------------------------------------------------------- carriers
def _redis_module():
    try:
        import redis
        return redis.StrictRedis, redis.ConnectionError
    except ImportError:
        from . import resp
        return resp.Client, resp.ConnectionError


def _server_cfg(redis_cfg):
    return {k: v for k, v in redis_cfg.items() if k != 'transport'}


class _Pending(Exception):
    """raised by an attempt that should be made again"""


def _bounded_retry(attempt, what, tries, base_delay):
    """Call attempt() until it returns; an attempt that raises _Pending is repeated after a pause, at most `tries` attempts in
    all.  The pause is per process -- between 1x and 2x base_delay, spread by pid -- so that a fleet of workers started
    together does not knock on the server in step (the behaviour, not the code, of dist.py:27-64).  -> attempt()'s value,
    or None when every attempt was pending."""
    pause = base_delay * (9 + os.getpid() % 10) / 9.0
    for n in range(1, tries + 1):
        try:
            return attempt()
        except _Pending as why:
            if n == tries:
                return None
            logger.warning('%s: %s; attempt %d of %d in %.2f s', what, why, n + 1, tries, pause)
            time.sleep(pause)


def retry_connect(redis_cfg, tries=300, base_delay=4., connect_timeout=None):
    """A connection that answered PING, or the carrier's ConnectionError after `tries` refusals (dist.py:27-43)."""
    client, refused = _redis_module()
    kw = _server_cfg(redis_cfg)
    if connect_timeout is not None:
        kw['socket_connect_timeout'] = connect_timeout
    last = []

    def attempt():
        try:
            conn = client(**kw)
            conn.ping()
            return conn
        except refused as e:
            last[:] = [e]
            raise _Pending('no answer from {} ({})'.format(redis_cfg, e))
    conn = _bounded_retry(attempt, 'connect', tries, base_delay)
    if conn is None:
        raise last[0]
    return conn


def retry_get(r, key, tries=300, base_delay=4.):
    """The value of `key` (GET), or of every key of a list / tuple (MGET), once all of them are set; RuntimeError after `tries`
    looks (dist.py:46-64)."""
    many = isinstance(key, (list, tuple))

    def attempt():
        got = r.mget(key) if many else [r.get(key)]
        if any(v is None for v in got):
            raise _Pending('{} not set'.format(key))
        return got if many else got[0]
    val = _bounded_retry(attempt, 'get', tries, base_delay)
    if val is None:
        raise RuntimeError('{} not set'.format(key))
    return val


def _carrier(redis_cfg):
    """-> ('redis', connection) or ('inprocess', broker)"""
    if not isinstance(redis_cfg, dict):
        return 'inprocess', _broker(redis_cfg)
    names_server = any(redis_cfg.get(k) for k in ('host', 'unix_socket_path', 'port'))
    mode = redis_cfg.get('transport') or os.environ.get('DNE_TRANSPORT') or ('redis' if names_server else 'inprocess')
    if mode == 'inprocess':
        return 'inprocess', _broker(redis_cfg)
    if mode == 'redis':
        return 'redis', retry_connect(redis_cfg)
    if mode != 'auto':
        raise ValueError('unknown transport {!r} (redis, inprocess, auto)'.format(mode))
    _, conn_err = _redis_module()
    try:                                  # auto: one attempt, then the in-process broker
        return 'redis', retry_connect(redis_cfg, tries=1, connect_timeout=2.0)
    except (conn_err, ConnectionError, OSError) as e:
        logger.warning('no Redis at {} ({}): transport "auto" falls back to the in-process broker -- only threads of THIS '
                       'process can meet there'.format(redis_cfg, e))
        return 'inprocess', _broker(redis_cfg)


def _wait_inprocess(cv, ready, what):
    """Block on the in-process broker until ready() -- called with cv held.  Nobody outside this process can ever satisfy the
    wait, so it is bounded (DNE_INPROCESS_TIMEOUT seconds, default 1200: 300 tries x 4 s of dist.py:46-64) and ends in the
    reference's error instead of a silent hang."""
    deadline = time.time() + float(os.environ.get('DNE_INPROCESS_TIMEOUT', '1200'))
    while not ready():
        left = deadline - time.time()
        if left <= 0:
            raise RuntimeError('{} not set (in-process broker: no master thread in this process declared it)'.format(what))
        cv.wait(min(left, 5.0))


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
    key = repr(sorted(_server_cfg(redis_cfg).items())) if isinstance(redis_cfg, dict) else repr(redis_cfg)
    with _brokers_lock:
        if key not in _brokers:
            _brokers[key] = _Broker()
        return _brokers[key]


def reset_brokers():
    with _brokers_lock:
        _brokers.clear()


# ---------------------------------------------------------------------------------------------- clients
class MasterClient:
    """dist.py:62-98"""

    def __init__(self, master_redis_cfg):
        self.task_counter = 0
        self.kind, self.r = _carrier(master_redis_cfg)
        logger.info('[master] transport: {}'.format(self.kind))

    def declare_experiment(self, exp):
        if self.kind == 'redis':
            self.r.set(EXP_KEY, serialize(exp))
            return
        with self.r.cv:
            self.r.exp = exp
            self.r.cv.notify_all()

    def declare_task(self, task_data):
        task_id = self.task_counter
        self.task_counter += 1
        if self.kind == 'redis':   # dist.py:76-80 (the reference pipelines the two commands; order is what matters)
            blob = serialize(task_data)
            self.r.mset({TASK_ID_KEY: task_id, TASK_DATA_KEY: blob})
            self.r.publish(TASK_CHANNEL, serialize((task_id, blob)))
            return task_id
        with self.r.cv:
            self.r.task_id, self.r.task_data = task_id, task_data
            self.r.cv.notify_all()
        return task_id

    def pop_result(self):
        if self.kind == 'redis':
            task_id, result = deserialize(self.r.blpop(RESULTS_KEY)[1])
            return task_id, result
        with self.r.cv:
            while not self.r.results:
                self.r.cv.wait()
            return self.r.results.popleft()

    def flush_results(self):
        if self.kind == 'redis':   # dist.py:90-91: everything but the newest entry
            n = self.r.llen(RESULTS_KEY)
            self.r.ltrim(RESULTS_KEY, -1, -1)
            return max(n - 1, 0)
        with self.r.cv:
            n = len(self.r.results)
            self.r.results.clear()
            return n

    def add_to_novelty_archive(self, novelty_vector):
        if self.kind == 'redis':
            self.r.rpush(ARCHIVE_KEY, serialize(novelty_vector))
            return
        with self.r.cv:
            self.r.archive.append(novelty_vector)

    def get_archive(self):
        if self.kind == 'redis':
            return [deserialize(v) for v in self.r.lrange(ARCHIVE_KEY, 0, -1)]
        with self.r.cv:
            return list(self.r.archive)


class RelayClient:
    """dist.py:101-151: one per worker machine -- mirrors the master's experiment and task into the local Redis and
    forwards the workers' results in batches.  Redis only (the in-process broker needs no relay)."""

    def __init__(self, master_redis_cfg, relay_redis_cfg):
        self.master_redis = retry_connect(master_redis_cfg)
        self.local_redis = retry_connect(relay_redis_cfg)
        self.results_published = 0
        self._stop = threading.Event()
        self._task_lock, self._last_task_id = threading.Lock(), -1

    def run(self, max_batches=None):
        self.local_redis.set(EXP_KEY, retry_get(self.master_redis, EXP_KEY))
        # dist.py:113-121 reads the current task and then subscribes: a task the master declares in between is lost until the
        # next one.  Here the subscription comes first and task ids only ever move forward locally, so nothing is missed and an
        # older task never overwrites a newer one.
        handler = lambda data: self._declare_task_local(*deserialize(data))   # noqa: E731
        if hasattr(self.master_redis, 'subscribe_loop'):
            ready = threading.Event()
            t = threading.Thread(target=self.master_redis.subscribe_loop, args=(TASK_CHANNEL, handler, self._stop, ready), daemon=True)
            t.start()
            ready.wait(30)
        else:                                                                  # the real redis package
            p = self.master_redis.pubsub(ignore_subscribe_messages=True)
            p.subscribe(**{TASK_CHANNEL: lambda msg: handler(msg['data'])})
            p.run_in_thread(sleep_time=0.001)
        self._declare_task_local(*retry_get(self.master_redis, (TASK_ID_KEY, TASK_DATA_KEY)))
        # forward worker results to the master in batches (dist.py:127-133 collects for a millisecond): block for one result, then take
        # what has queued up behind it in the meantime -- a burst of workers finishing together costs the master one RPUSH, a lone
        # result is forwarded at once.  This relay is not the list's only consumer: flush_results (a new task, on the subscription
        # thread) trims it, so the follow-up pops carry a timeout and the batch in hand is forwarded as soon as one comes back empty --
        # the loop never sits in an unbounded pop while holding results
        batches = 0
        while max_batches is None or batches < max_batches:
            batch = [self.local_redis.blpop(RESULTS_KEY)[1]]
            for _ in range(min(int(self.local_redis.llen(RESULTS_KEY)), 4096)):
                more = self.local_redis.blpop(RESULTS_KEY, timeout=1)
                if more is None:
                    break
                batch.append(more[1])
            self.master_redis.rpush(RESULTS_KEY, *batch)
            self.results_published += len(batch)
            batches += 1

    def flush_results(self):
        for r in (self.local_redis, self.master_redis):
            r.llen(RESULTS_KEY)
            r.ltrim(RESULTS_KEY, -1, -1)

    def _declare_task_local(self, task_id, task_data):
        if isinstance(task_id, bytes):
            task_id = int(task_id)
        with self._task_lock:
            if task_id <= self._last_task_id:      # the subscription already delivered this task or a newer one
                return
            self._last_task_id = task_id
            logger.info('[relay] Received task {}'.format(task_id))
            self.results_published = 0
            self.local_redis.mset({TASK_ID_KEY: task_id, TASK_DATA_KEY: task_data})
            self.flush_results()


class WorkerClient:
    """dist.py:153-192.  Argument order as in the reference class: (relay_redis_cfg, master_redis_cfg)."""

    def __init__(self, relay_redis_cfg, master_redis_cfg):
        # a GPU worker shares the master's host more often than not: with the in-process broker, and whenever no relay
        # is running, both names mean the master's carrier
        self.kind, self.master = _carrier(master_redis_cfg)
        if self.kind == 'redis':
            try:
                self.local = retry_connect(relay_redis_cfg, tries=1, connect_timeout=2.0) if relay_redis_cfg != master_redis_cfg else self.master
            except Exception:
                logger.warning('[worker] no relay at {}: talking to the master directly'.format(relay_redis_cfg))
                self.local = self.master
        else:
            self.local = self.master
        self.cached_task_id, self.cached_task_data = None, None

    def get_experiment(self):
        if self.kind == 'redis':
            return deserialize(retry_get(self.local, EXP_KEY))
        with self.master.cv:
            _wait_inprocess(self.master.cv, lambda: self.master.exp is not None, EXP_KEY)
            return self.master.exp

    def get_archive(self):
        if self.kind == 'redis':   # append-only list (dist.py:93-95): fetch and unpickle only what is new
            if not hasattr(self, '_archive'):
                self._archive = []
            self._archive += [deserialize(v) for v in self.master.lrange(ARCHIVE_KEY, len(self._archive), -1)]
            return list(self._archive)
        with self.master.cv:
            return list(self.master.archive)

    def get_current_task(self):
        if self.kind == 'redis':
            # dist.py:172-188 reads the id under WATCH and fetches the data only when it changed; MGET returns both keys
            # atomically, so one extra check of the id is enough
            task_id = int(retry_get(self.local, TASK_ID_KEY))
            while task_id != self.cached_task_id:
                tid, blob = retry_get(self.local, (TASK_ID_KEY, TASK_DATA_KEY))
                self.cached_task_id, self.cached_task_data = int(tid), deserialize(blob)
                task_id = int(retry_get(self.local, TASK_ID_KEY))
            return self.cached_task_id, self.cached_task_data
        with self.master.cv:
            _wait_inprocess(self.master.cv, lambda: self.master.task_data is not None, TASK_DATA_KEY)
            return self.master.task_id, self.master.task_data

    def push_result(self, task_id, result):
        if self.kind == 'redis':
            self.local.rpush(RESULTS_KEY, serialize((task_id, result)))
            return
        with self.master.cv:
            self.master.results.append((task_id, result))
            self.master.cv.notify_all()
