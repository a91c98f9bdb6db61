"""TEST-ONLY: a tiny Redis-protocol server (RESP2 over TCP or a unix socket) with just the commands the
es_distributed transport uses, so that the Redis carrier of dne_hip/dist.py -- framing, keys, pickled payloads,
blocking pops, publish/subscribe -- can be exercised in a container that has no Redis."""
import os
import socket
import threading
import time
from collections import defaultdict, deque


class FakeRedis:
    def __init__(self, unix_path=None):
        self.kv, self.lists, self.subs = {}, defaultdict(deque), defaultdict(list)
        self.cv = threading.Condition()
        self.commands = []
        if unix_path:
            if os.path.exists(unix_path):
                os.unlink(unix_path)
            self.srv = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
            self.srv.bind(unix_path)
            self.cfg = {"unix_socket_path": unix_path}
        else:
            self.srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            self.srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            self.srv.bind(("127.0.0.1", 0))
            self.cfg = {"host": "127.0.0.1", "port": self.srv.getsockname()[1]}
        self.srv.listen(64)
        self.alive = True
        threading.Thread(target=self._accept, daemon=True).start()

    def close(self):
        self.alive = False
        try:
            self.srv.close()
        except OSError:
            pass

    def _accept(self):
        while self.alive:
            try:
                c, _ = self.srv.accept()
            except OSError:
                return
            threading.Thread(target=self._serve, args=(c,), daemon=True).start()

    @staticmethod
    def _reader(c):
        buf = b""

        def line():
            nonlocal buf
            while b"\r\n" not in buf:
                d = c.recv(1 << 20)
                if not d:
                    raise EOFError
                buf += d
            l, buf = buf.split(b"\r\n", 1)
            return l

        def exact(n):
            nonlocal buf
            while len(buf) < n + 2:
                d = c.recv(1 << 20)
                if not d:
                    raise EOFError
                buf += d
            d, buf = buf[:n], buf[n + 2:]
            return d

        def command():
            l = line()
            assert l[:1] == b"*", l
            return [exact(int(line()[1:])) for _ in range(int(l[1:]))]
        return command

    @staticmethod
    def _bulk(v):
        return b"$-1\r\n" if v is None else b"$%d\r\n%s\r\n" % (len(v), v)

    def _serve(self, c):
        read = self._reader(c)
        try:
            while True:
                a = read()
                op = a[0].upper()
                self.commands.append(op)
                if op == b"PING":
                    c.sendall(b"+PONG\r\n")
                elif op == b"SET":
                    with self.cv:
                        self.kv[a[1]] = a[2]
                    c.sendall(b"+OK\r\n")
                elif op == b"GET":
                    with self.cv:
                        v = self.kv.get(a[1])
                    c.sendall(self._bulk(v))
                elif op == b"MSET":
                    with self.cv:
                        for i in range(1, len(a), 2):
                            self.kv[a[i]] = a[i + 1]
                    c.sendall(b"+OK\r\n")
                elif op == b"MGET":
                    with self.cv:
                        vs = [self.kv.get(k) for k in a[1:]]
                    c.sendall(b"*%d\r\n" % len(vs) + b"".join(self._bulk(v) for v in vs))
                elif op == b"RPUSH":
                    with self.cv:
                        self.lists[a[1]].extend(a[2:])
                        n = len(self.lists[a[1]])
                        self.cv.notify_all()
                    c.sendall(b":%d\r\n" % n)
                elif op == b"BLPOP":
                    tmo = float(a[2]) if len(a) > 2 else 0.0   # BLPOP key timeout: 0 = for ever, else a nil reply when it runs out (Redis)
                    end = time.time() + tmo
                    v = None
                    with self.cv:
                        while not self.lists[a[1]]:
                            left = end - time.time()
                            if tmo > 0 and left <= 0:
                                break
                            self.cv.wait(left if tmo > 0 else None)
                        if self.lists[a[1]]:
                            v = self.lists[a[1]].popleft()
                    c.sendall(b"*-1\r\n" if v is None else b"*2\r\n" + self._bulk(a[1]) + self._bulk(v))
                elif op == b"LLEN":
                    with self.cv:
                        n = len(self.lists[a[1]])
                    c.sendall(b":%d\r\n" % n)
                elif op == b"LTRIM":
                    with self.cv:
                        l = list(self.lists[a[1]])
                        s, e = int(a[2]), int(a[3])
                        s = max(len(l) + s, 0) if s < 0 else s
                        e = len(l) + e if e < 0 else e
                        self.lists[a[1]] = deque(l[s:e + 1])
                    c.sendall(b"+OK\r\n")
                elif op == b"LRANGE":
                    with self.cv:
                        l = list(self.lists[a[1]])
                    s, e = int(a[2]), int(a[3])
                    e = len(l) + e if e < 0 else e
                    vs = l[s:e + 1]
                    c.sendall(b"*%d\r\n" % len(vs) + b"".join(self._bulk(v) for v in vs))
                elif op == b"PUBLISH":
                    with self.cv:
                        targets = list(self.subs[a[1]])
                    for t in targets:
                        try:
                            t.sendall(b"*3\r\n" + self._bulk(b"message") + self._bulk(a[1]) + self._bulk(a[2]))
                        except OSError:
                            pass
                    c.sendall(b":%d\r\n" % len(targets))
                elif op == b"SUBSCRIBE":
                    with self.cv:
                        self.subs[a[1]].append(c)
                    c.sendall(b"*3\r\n" + self._bulk(b"subscribe") + self._bulk(a[1]) + b":1\r\n")
                else:
                    c.sendall(b"-ERR unknown command\r\n")
        except (EOFError, OSError, AssertionError):
            pass
        finally:
            c.close()
