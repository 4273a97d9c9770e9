"""storage.RemoteCache over the engine, and the sets as a Redis protocol stream.

`GpuRemoteCache` is the drop-in for RedisCache on this path (storage/rediscache.go; interface storage/types.go:83-102):
`serials::…` sets live in the HBM table behind libctmr, every other key (crl::, issuer::, log state) in the library's
host-side store.  `redis_dump` writes the sets as RESP commands that `redis-cli --pipe` loads into the Redis of a
reference deployment, so that a real ct-fetch run elsewhere and this engine can be diffed with the reference's own
tools (storage-statistics, SCARD/SMEMBERS); `redis_load` is the way back.  (The Python twin of the rest of the
reference's storage package is test scaffolding: tests/storage_mirror.py; the host mirror a cgo binding would follow is
include/ctmr_storage.hpp.)"""
import calendar

from .engine import Engine


def _b(x):
    return x.encode() if isinstance(x, str) else bytes(x)


class GpuRemoteCache:
    def __init__(self, engine: Engine):
        self.engine = engine

    def SetInsert(self, key, entry): return self.engine.set_insert(_b(key), _b(entry))
    def SetRemove(self, key, entry): return self.engine.set_remove(_b(key), _b(entry))
    def SetContains(self, key, entry): return self.engine.set_contains(_b(key), _b(entry))
    def SetList(self, key): return self.engine.set_list(_b(key))
    def SetToChan(self, key): return iter(self.engine.set_list(_b(key)))
    def SetCardinality(self, key): return self.engine.set_cardinality(_b(key))
    def Exists(self, key): return self.engine.exists(_b(key))
    def ExpireAt(self, key, unix_seconds): self.engine.expire_at(_b(key), unix_seconds)
    def KeysToChan(self, pattern): return iter(self.engine.keys(_b(pattern)))

    def StoreLogStateJSON(self, shortURL, doc: bytes):       # rediscache.go:180-190: key "log::<shortURL>", one JSON document
        key = b"log::" + _b(shortURL)
        for old in self.engine.set_list(key):
            self.engine.set_remove(key, old)
        self.engine.set_insert(key, doc)

    def LoadLogStateJSON(self, shortURL) -> bytes:
        d = self.engine.set_list(b"log::" + _b(shortURL))
        if not d:
            raise KeyError("Log state not found")
        return d[0]


_SET_PATTERNS = ("serials::*", "crl::*", "issuer::*")


def exp_date_expire_time(exp_date_id: str) -> int:
    """ExpDate.ExpireTime of NewExpDate(id): the start of the hour ("2006-01-02-15") or day ("2006-01-02"), unix seconds
    (storage/types.go:348-367,398-400)."""
    parts = [int(x) for x in exp_date_id.split("-")]
    if len(parts) not in (3, 4):
        raise ValueError("not an ExpDate id: %r" % exp_date_id)
    y, mo, d = parts[:3]
    return calendar.timegm((y, mo, d, parts[3] if len(parts) == 4 else 0, 0, 0))


def _resp(*args) -> bytes:
    out = [b"*%d\r\n" % len(args)]
    for a in args:
        a = _b(a)
        out.append(b"$%d\r\n" % len(a) + a + b"\r\n")
    return b"".join(out)


def redis_dump(cache, out, patterns=_SET_PATTERNS, members_per_command=512) -> dict:
    """Writes SADD commands for every set matching `patterns` (members are raw bytes — serials contain NULs, which
    RESP bulk strings carry unchanged) and, for the known-certificate sets, the EXPIREAT the reference puts on them:
    the expDate of the key (KnownCertificates.setExpiryFlag, storage/knowncertificates.go:98-104).  `cache`: anything
    with KeysToChan / SetToChan (GpuRemoteCache, a mock).  → counts."""
    n_keys = n_members = 0
    prefix = b"serials::"
    for pat in patterns:
        for key in sorted(cache.KeysToChan(pat)):
            key = _b(key)
            members = sorted(set(cache.SetToChan(key)))       # SetToChan may repeat members (knowncertificates.go:80-93)
            for i in range(0, len(members), members_per_command):
                out.write(_resp(b"SADD", key, *members[i:i + members_per_command]))
            if key.startswith(prefix):
                out.write(_resp(b"EXPIREAT", key, str(exp_date_expire_time(key[len(prefix):].split(b"::", 1)[0].decode()))))
            n_keys += 1
            n_members += len(members)
    return {"keys": n_keys, "members": n_members}


def redis_load(cache, stream) -> dict:
    """Applies a redis_dump() stream (RESP arrays of bulk strings; SADD and EXPIREAT) to `cache`."""
    data = stream.read()
    pos, n_cmd, n_new = 0, 0, 0

    def line():
        nonlocal pos
        e = data.index(b"\r\n", pos)
        v = data[pos:e]
        pos = e + 2
        return v

    while pos < len(data):
        head = line()
        if head[:1] != b"*":
            raise ValueError("not a RESP array at byte %d" % (pos - len(head) - 2))
        args = []
        for _ in range(int(head[1:])):
            ln = line()
            if ln[:1] != b"$":
                raise ValueError("not a bulk string at byte %d" % (pos - len(ln) - 2))
            n = int(ln[1:])
            args.append(data[pos:pos + n])
            if data[pos + n:pos + n + 2] != b"\r\n":
                raise ValueError("bulk string not terminated at byte %d" % (pos + n))
            pos += n + 2
        cmd = args[0].upper()
        if cmd == b"SADD":
            for m in args[2:]:
                n_new += bool(cache.SetInsert(args[1], m))
        elif cmd == b"EXPIREAT":
            cache.ExpireAt(args[1], int(args[2]))
        else:
            raise ValueError("unsupported command %r" % cmd)
        n_cmd += 1
    return {"commands": n_cmd, "inserted": n_new}
