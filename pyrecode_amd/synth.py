"""Host (numpy) mirror of the device synthetic-stack generator in csrc/rc_expand.hip: same integers."""
import numpy as np

_M1, _M2 = np.uint32(0x7FEB352D), np.uint32(0x846CA68B)


def _mix32(x):
    x = x.astype(np.uint32, copy=True)
    x ^= x >> np.uint32(16)
    x *= _M1
    x ^= x >> np.uint32(15)
    x *= _M2
    x ^= x >> np.uint32(16)
    return x


def _mix32s(v):
    return int(_mix32(np.array([v & 0xFFFFFFFF], np.uint32))[0])


def dark_frame(seed, n_pixels):
    i = np.arange(n_pixels, dtype=np.uint32)
    key = np.uint32(_mix32s(seed ^ 0xD1B54A32))
    return (np.uint32(80) + _mix32(i ^ key) % np.uint32(41)).astype(np.uint16)


def frames(seed, first_frame, n_frames, n_pixels, sparsity_ppm, dark):
    thresh24 = np.uint32((sparsity_ppm << 24) // 1000000)
    i = np.arange(n_pixels, dtype=np.uint32)
    dk = dark.astype(np.uint32).ravel()
    out = np.empty((n_frames, n_pixels), np.uint16)
    for z in range(n_frames):
        fkey = np.uint32(_mix32s((seed + 0x9E3779B9 * (first_frame + z + 1)) & 0xFFFFFFFF))
        h = _mix32(i ^ fkey)
        h2 = _mix32(h ^ np.uint32(0x68E31DA4))
        ev = (h & np.uint32(0xFFFFFF)) < thresh24
        out[z] = np.where(ev, dk + np.uint32(1) + h2 % np.uint32(2047), h2 % (dk + np.uint32(1))).astype(np.uint16)
    return out


def frames_clustered(seed, first_frame, n_frames, nx, ny, seed_ppm, dark):
    """Host mirror of rc_synth_frames_clustered (csrc/rc_expand.hip::k_synth_frames_clustered): events in clusters of 1..6
    pixels inside the 2 x 3 window a seed pixel anchors (top-left)."""
    thresh24 = np.uint32((seed_ppm << 24) // 1000000)
    n = nx * ny
    i = np.arange(n, dtype=np.uint32)
    dk = dark.astype(np.uint32).ravel()
    out = np.empty((n_frames, n), np.uint16)
    for z in range(n_frames):
        fkey = np.uint32(_mix32s((seed + 0x9E3779B9 * (first_frame + z + 1)) & 0xFFFFFFFF))
        hq = _mix32(i ^ fkey)
        is_seed = ((hq & np.uint32(0xFFFFFF)) < thresh24).reshape(ny, nx)
        shape = _mix32(hq ^ np.uint32(0x3C6EF372)).reshape(ny, nx)
        ev = np.zeros((ny, nx), bool)
        for dy in range(2):
            for dx in range(3):
                cell = dy * 3 + dx
                lit = is_seed if cell == 0 else is_seed & (((shape >> np.uint32(3 * cell)) & np.uint32(7)) < 5)
                ev[dy:, dx:] |= lit[:ny - dy, :nx - dx]
        h2 = _mix32(hq ^ np.uint32(0x68E31DA4))
        out[z] = np.where(ev.ravel(), dk + np.uint32(1) + h2 % np.uint32(2047), h2 % (dk + np.uint32(1))).astype(np.uint16)
    return out
