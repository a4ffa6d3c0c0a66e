"""Serial restatement (test infrastructure) of the two LZ4 block parses the device encoder implements
(pyrecode_amd/csrc/rc_lz4_block.h): lz4_parse_runs (compression_level 0) and lz4_parse_events (>= 1).  Used two ways:
on the CPU stock liblz4 must decode what the model emits (the PARSE is format-conformant, and its ratio is what DESIGN.md
claims); on the GPU the device's bytes must equal the model's, block for block."""
import numpy as np

EV_MAX = 62      # one event per lane (lane 0 = the block start)
EV_MAX2 = 126    # two events per lane (rc_lz4_block.h::lz4_parse_events2): the same rules, event numbers take 7 bits of the keys
NM_MAX2 = 126    # ... and at most this many matches (the position tables' size); beyond either the run parser
NZ_STORE = 272   # compression_level >= 1: a block with more non-zero bytes than this is stored without a parse (rc_lz4_block.h::LZ4_NZ_STORE)


def emit(block, matches):
    """LZ4 block bytes of `block` with the given non-overlapping, position-ordered (start, length, offset) matches."""
    n, out, pos = len(block), bytearray(), 0

    def ext(v):
        while v >= 255:
            out.append(255)
            v -= 255
        out.append(v)
    for s, l, off in matches:
        ll = s - pos
        out.append((min(ll, 15) << 4) | min(l - 4, 15))
        if ll >= 15:
            ext(ll - 15)
        out += block[pos:s]
        out += bytes([off & 255, off >> 8])
        if l - 4 >= 15:
            ext(l - 4 - 15)
        pos = s + l
    ll = n - pos
    out.append(min(ll, 15) << 4)
    if ll >= 15:
        ext(ll - 15)
    out += block[pos:]
    return bytes(out)


def parse_runs(block):
    """every run of >= 5 zero bytes that lies in front of the block's last 12 bytes = [literal 00][match offset 1]"""
    n = len(block)
    z = np.frombuffer(block, np.uint8) == 0
    z[max(n - 12, 0):] = False
    matches, a = [], None
    for i in range(n + 1):
        if i < n and z[i]:
            if a is None:
                a = i
        else:
            if a is not None and i - a >= 5:
                matches.append((a + 1, i - a - 1, 1))
            a = None
    return matches


def parse_events(block):
    """rc_lz4_block.h::lz4_parse_events / lz4_parse_events2, event by event.  None: more than EV_MAX2 events or NM_MAX2 matches
    (the device takes the run parser)."""
    n = len(block)
    b = np.frombuffer(block, np.uint8)
    ev = np.nonzero(b)[0].tolist()
    if len(ev) > EV_MAX2:
        return None
    LB = 6 if len(ev) <= EV_MAX else 7                   # bits of the event number inside a key
    P1 = [0] + [p + 1 for p in ev]                       # lane 0 = the block start
    K = len(P1)
    Pn = P1[1:] + [n + 1]
    R = [Pn[k] - P1[k] - 1 for k in range(K)]
    cls = [None] + [(int(b[p]).bit_length() - 1) if (b[p] & (b[p] - 1)) == 0 else None for p in ev]
    best = {}                                            # value class -> (key, lane) of the best earlier event
    has, trail, lead, J = [False] * K, [0] * K, [0] * K, [0] * K
    zbest, zb = [0] * K, 0                               # key of the longest zero run among the lanes in front: min(R, 511) << 6 | lane
    for k in range(K):
        zbest[k] = zb
        zb = max(zb, (min(R[k], 511) << LB) | k)
        c = cls[k]
        if c is not None and c in best:
            j = best[c][1]
            t = min(R[k], R[j], n - 5 - P1[k])
            if P1[k] + 11 <= n and t >= 3:
                has[k], trail[k], J[k] = True, t, j
        if c is not None:
            key = (min(R[k], 127) << LB) | k
            if c not in best or key > best[c][0]:
                best[c] = (key, k)
    d = [R[k] - trail[k] for k in range(K)]
    for k in range(1, K):
        if has[k]:
            lead[k] = min(d[k - 1], R[J[k] - 1])
    matches = []
    for k in range(K):
        if has[k]:
            matches.append((P1[k] - 1 - lead[k], lead[k] + 1 + trail[k], P1[k] - P1[J[k]]))
        leadn = lead[k + 1] if k + 1 < K else 0
        gs = P1[k] + trail[k] + (0 if has[k] else 1)
        ge = min(Pn[k] - 1 - leadn, n - 5)
        off = 1
        L0 = ge - P1[k]
        if k > 0 and not has[k] and L0 >= 4 and (zbest[k] >> LB) >= L0 and P1[k] + 12 <= n:
            # no unit copy for this event: the gap behind X comes from inside the longest earlier zero run (no literal zero needed)
            gs, off = P1[k], P1[k] - P1[zbest[k] & ((1 << LB) - 1)]
        if gs + 12 <= n and ge - gs >= 4:
            matches.append((gs, ge - gs, off))
    if LB == 7 and len(matches) > NM_MAX2:
        return None
    return matches


def encode_block(block, level):
    """(size word, payload) as they stand in an LZ4 frame: the compressed block, or the block stored (bit 31) if that is not smaller"""
    if level and int(np.count_nonzero(np.frombuffer(block, np.uint8))) > NZ_STORE:
        return len(block) | 0x80000000, bytes(block)
    m = parse_events(block) if level else None
    if m is None:
        m = parse_runs(block)
    enc = emit(block, m)
    if len(enc) >= len(block):
        return len(block) | 0x80000000, bytes(block)
    return len(enc), enc


def frame_blocks(frame):
    """[(size word, payload)] of an LZ4 frame with a 7-byte header and no block checksums"""
    out, q = [], 7
    while True:
        w = int.from_bytes(frame[q:q + 4], "little")
        q += 4
        if w == 0:
            break
        size = w & 0x7FFFFFFF
        out.append((w, bytes(frame[q:q + size])))
        q += size
    assert q == len(frame)
    return out
