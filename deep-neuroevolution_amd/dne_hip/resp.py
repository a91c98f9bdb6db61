"""A small Redis client (RESP2 over a unix or TCP socket) -- just the commands the es_distributed transport uses
(dist.py:62-192: SET GET MGET MSET PUBLISH SUBSCRIBE BLPOP RPUSH LLEN LTRIM LRANGE PING).

The reference talks to Redis through the `redis` package, which is not part of this image; the wire protocol is
simple enough to speak directly, so HIP workers can join a reference master's Redis (and the reverse) without a new
dependency.  dist.py prefers the real `redis` package when it is importable and uses this client otherwise.
"""
import socket
import threading


class RespError(RuntimeError):
    pass


class ConnectionError(OSError):   # the name the reference catches (redis.ConnectionError, dist.py:33)
    pass


def _enc(x):
    if isinstance(x, bytes):
        return x
    if isinstance(x, (bytearray, memoryview)):
        return bytes(x)
    return str(x).encode()


class Connection:
    """One socket, one command in flight.  redis_cfg as in the reference: {'unix_socket_path': p} or
    {'host': h, 'port': n} (main.py:56,76-77)."""

    def __init__(self, redis_cfg, timeout=None):
        try:
            if redis_cfg.get("unix_socket_path"):
                self.sock = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
                self.sock.settimeout(timeout)
                self.sock.connect(redis_cfg["unix_socket_path"])
            else:
                self.sock = socket.create_connection((redis_cfg.get("host", "127.0.0.1"), int(redis_cfg.get("port", 6379))),
                                                     timeout=timeout)
                self.sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
        except OSError as e:
            raise ConnectionError("cannot connect to redis at %r: %s" % (redis_cfg, e))
        self.sock.settimeout(None)
        self.buf = b""

    def close(self):
        try:
            self.sock.close()
        except OSError:
            pass

    # ---- protocol
    def send(self, *args):
        out = [b"*%d\r\n" % len(args)]
        for a in args:
            a = _enc(a)
            out.append(b"$%d\r\n" % len(a))
            out.append(a)
            out.append(b"\r\n")
        self.sock.sendall(b"".join(out))

    def _fill(self):
        chunk = self.sock.recv(1 << 20)
        if not chunk:
            raise ConnectionError("redis closed the connection")
        self.buf += chunk

    def _line(self):
        while True:
            i = self.buf.find(b"\r\n")
            if i >= 0:
                line, self.buf = self.buf[:i], self.buf[i + 2:]
                return line
            self._fill()

    def _exact(self, n):
        while len(self.buf) < n + 2:
            self._fill()
        data, self.buf = self.buf[:n], self.buf[n + 2:]
        return data

    def read(self):
        line = self._line()
        t, rest = line[:1], line[1:]
        if t == b"+":
            return rest
        if t == b"-":
            raise RespError(rest.decode(errors="replace"))
        if t == b":":
            return int(rest)
        if t == b"$":
            n = int(rest)
            return None if n < 0 else self._exact(n)
        if t == b"*":
            n = int(rest)
            return None if n < 0 else [self.read() for _ in range(n)]
        raise RespError("bad reply type %r" % line[:20])

    def call(self, *args):
        self.send(*args)
        return self.read()


class _Locked:
    """a connection shared by threads: one command (request + reply) at a time"""

    def __init__(self, conn):
        self.conn, self.lock = conn, threading.Lock()

    def call(self, *args):
        with self.lock:
            return self.conn.call(*args)


class Client:
    """The subset of redis.StrictRedis the transport needs, same method names and return shapes."""

    def __init__(self, socket_connect_timeout=None, **redis_cfg):
        self.cfg = dict(redis_cfg)
        self.timeout = socket_connect_timeout
        self.c = _Locked(Connection(self.cfg, timeout=socket_connect_timeout))
        self._blocking = None     # BLPOP gets its own connection: it may sit there while other threads keep talking

    def ping(self):
        return self.c.call("PING") == b"PONG"

    def set(self, key, value):
        return self.c.call("SET", key, value) == b"OK"

    def get(self, key):
        return self.c.call("GET", key)

    def mget(self, keys):
        return self.c.call("MGET", *keys)

    def mset(self, mapping):
        flat = []
        for k, v in mapping.items():
            flat += [k, v]
        return self.c.call("MSET", *flat) == b"OK"

    def publish(self, channel, message):
        return self.c.call("PUBLISH", channel, message)

    def rpush(self, key, *values):
        return self.c.call("RPUSH", key, *values)

    def blpop(self, key, timeout=0):
        if self._blocking is None:
            self._blocking = _Locked(Connection(self.cfg, timeout=self.timeout))
        r = self._blocking.call("BLPOP", key, timeout)
        return None if r is None else (r[0], r[1])

    def llen(self, key):
        return self.c.call("LLEN", key)

    def ltrim(self, key, start, stop):
        return self.c.call("LTRIM", key, start, stop) == b"OK"

    def lrange(self, key, start, stop):
        return self.c.call("LRANGE", key, start, stop)

    def subscribe_loop(self, channel, handler, stop_event=None, ready_event=None):
        """Blocking: a dedicated connection receives `channel`'s messages and passes each payload to handler.  ready_event is set
        once the server has confirmed the subscription (nothing published after that is missed)."""
        sub = Connection(self.cfg, timeout=self.timeout)
        sub.send("SUBSCRIBE", channel)
        try:
            while stop_event is None or not stop_event.is_set():
                msg = sub.read()
                if ready_event is not None and isinstance(msg, list) and len(msg) == 3 and msg[0] == b"subscribe":
                    ready_event.set()
                if isinstance(msg, list) and len(msg) == 3 and msg[0] == b"message":
                    handler(msg[2])
        finally:
            sub.close()
