"""Synthetic inputs for tests and bench.py: no dataset, mesh or checkpoint exists on disk, so the
workloads of BASELINE.json are restated as a deterministic procedural room.  numpy only; identical here and on the GPU box.

room(seed, n_tris): closed 4 x 3 x 2.6 m box whose six faces are regular grids displaced by smooth value noise (<= 1 cm),
plus 40 random boxes ("furniture") standing on the floor; emitters = ceiling triangles over a central 1 m x 1 m patch.
slf_for(...): H^3 voxel mask of the voxels the surface touches + a smooth positive radiance field (see SURVEY.md section 7 on
why the field is smooth).  camera(...): pinhole views from the room centre, OpenCV convention (utils/dataset/real_ldr.py).
"""
import numpy as np

ROOM = (4.0, 3.0, 2.6)


def _value_noise(p, rng_tab, freq=2.0):
    """Smooth trilinear value noise in [-1,1] at points p (N,3)."""
    q = p * freq
    i = np.floor(q).astype(np.int64)
    f = q - i
    f = f * f * (3 - 2 * f)
    n = rng_tab.shape[0]

    def h(ix, iy, iz):
        return rng_tab[(ix * 73856093 ^ iy * 19349663 ^ iz * 83492791) % n]
    out = 0
    for dx in (0, 1):
        for dy in (0, 1):
            for dz in (0, 1):
                w = (f[:, 0] if dx else 1 - f[:, 0]) * (f[:, 1] if dy else 1 - f[:, 1]) * (f[:, 2] if dz else 1 - f[:, 2])
                out = out + w * h(i[:, 0] + dx, i[:, 1] + dy, i[:, 2] + dz)
    return out


def _grid_quad(origin, eu, ev, nu, nv, flip=False):
    """(nu x nv)-cell grid on the parallelogram origin + s*eu + t*ev -> verts, faces, interior mask."""
    s = np.linspace(0, 1, nu + 1)
    t = np.linspace(0, 1, nv + 1)
    S, T = np.meshgrid(s, t, indexing="ij")
    v = origin[None, None] + S[..., None] * eu[None, None] + T[..., None] * ev[None, None]
    interior = np.ones((nu + 1, nv + 1), bool)
    interior[0] = interior[-1] = False
    interior[:, 0] = interior[:, -1] = False
    idx = np.arange((nu + 1) * (nv + 1)).reshape(nu + 1, nv + 1)
    a, b, c, d = idx[:-1, :-1], idx[1:, :-1], idx[1:, 1:], idx[:-1, 1:]
    f = np.concatenate([np.stack([a, b, c], -1).reshape(-1, 3), np.stack([a, c, d], -1).reshape(-1, 3)], 0)
    if flip:
        f = f[:, ::-1]
    return v.reshape(-1, 3), f, interior.reshape(-1)


def room(seed=0, n_tris=200_000, n_boxes=40):
    """Returns dict(vertices f32 (V,3), faces i32 (F,3), is_emitter bool (F,))."""
    rng = np.random.default_rng(seed)
    X, Y, Z = ROOM
    quads = []  # (origin, eu, ev, normal_inward, tag)
    quads.append((np.array([0, 0, 0.0]), np.array([X, 0, 0.0]), np.array([0, Y, 0.0]), np.array([0, 0, 1.0]), "floor"))
    quads.append((np.array([0, 0, Z]), np.array([X, 0, 0.0]), np.array([0, Y, 0.0]), np.array([0, 0, -1.0]), "ceiling"))
    quads.append((np.array([0, 0, 0.0]), np.array([X, 0, 0.0]), np.array([0, 0, Z]), np.array([0, 1.0, 0]), "wall"))
    quads.append((np.array([0, Y, 0.0]), np.array([X, 0, 0.0]), np.array([0, 0, Z]), np.array([0, -1.0, 0]), "wall"))
    quads.append((np.array([0, 0, 0.0]), np.array([0, Y, 0.0]), np.array([0, 0, Z]), np.array([1.0, 0, 0]), "wall"))
    quads.append((np.array([X, 0, 0.0]), np.array([0, Y, 0.0]), np.array([0, 0, Z]), np.array([-1.0, 0, 0]), "wall"))
    for _ in range(n_boxes):
        sx, sy, sz = rng.uniform(0.2, 1.0), rng.uniform(0.2, 0.8), rng.uniform(0.3, 1.6)
        cx, cy = rng.uniform(0.4, X - 0.4), rng.uniform(0.4, Y - 0.4)
        if abs(cx - X / 2) < 0.6 and abs(cy - Y / 2) < 0.6:   # keep the camera position free
            cx += 1.2 if cx < X / 2 else -1.2
        ang = rng.uniform(0, np.pi)
        ux = np.array([np.cos(ang), np.sin(ang), 0.0]) * sx
        uy = np.array([-np.sin(ang), np.cos(ang), 0.0]) * sy
        uz = np.array([0, 0, sz])
        o = np.array([cx, cy, 0.0]) - 0.5 * ux - 0.5 * uy
        nx, ny = ux / sx, uy / sy
        quads.append((o + uz, ux, uy, np.array([0, 0, 1.0]), "box"))
        quads.append((o, ux, uz, -ny, "box"))
        quads.append((o + uy, ux, uz, ny, "box"))
        quads.append((o, uy, uz, -nx, "box"))
        quads.append((o + ux, uy, uz, nx, "box"))
    area = sum(np.linalg.norm(np.cross(q[1], q[2])) for q in quads)
    h = np.sqrt(area / max(n_tris / 2.0, 1.0))
    noise_tab = rng.uniform(-1, 1, size=4096)
    V, F, E = [], [], []
    off = 0
    for o, eu, ev, nrm, tag in quads:
        nu = max(1, int(round(np.linalg.norm(eu) / h)))
        nv = max(1, int(round(np.linalg.norm(ev) / h)))
        v, f, interior = _grid_quad(o, eu, ev, nu, nv)
        disp = 0.01 * _value_noise(v, noise_tab, 2.5) * interior   # <= 1 cm, borders fixed so the shell stays closed
        v = v + disp[:, None] * nrm[None]
        cen = v[f].mean(1)
        if tag == "ceiling":
            em = (np.abs(cen[:, 0] - X / 2) < 0.5) & (np.abs(cen[:, 1] - Y / 2) < 0.5)
        else:
            em = np.zeros(len(f), bool)
        V.append(v); F.append(f + off); E.append(em)
        off += len(v)
    vertices = np.concatenate(V).astype(np.float32)
    faces = np.concatenate(F).astype(np.int32)
    return {"vertices": vertices, "faces": faces, "is_emitter": np.concatenate(E)}


def smooth_radiance(c):
    """0.25 + 0.2 sin(2 pi x/1.3) cos(2 pi y/1.7), phase-shifted per channel."""
    x, y, z = c[:, 0], c[:, 1], c[:, 2]
    r = 0.25 + 0.2 * np.sin(2 * np.pi * x / 1.3) * np.cos(2 * np.pi * y / 1.7)
    g = 0.25 + 0.2 * np.sin(2 * np.pi * y / 1.3 + 1.0) * np.cos(2 * np.pi * z / 1.7)
    b = 0.25 + 0.2 * np.sin(2 * np.pi * z / 1.3 + 2.0) * np.cos(2 * np.pi * x / 1.7)
    return np.stack([r, g, b], -1).astype(np.float32)


def slf_for(vertices, faces, H=256, voxel_min=-0.1, voxel_max=4.1):
    """vslf.npz contents for a mesh: mask (H,H,H) bool of voxels touched by the surface, inds int64, radiance (K,3)."""
    v = vertices.astype(np.float64)
    tri = v[faces]
    vox = (voxel_max - voxel_min) / H
    edge = np.linalg.norm(tri[:, 1] - tri[:, 0], axis=-1).max()
    edge = max(edge, np.linalg.norm(tri[:, 2] - tri[:, 0], axis=-1).max())
    n = int(min(64, max(2, np.ceil(edge / vox * 1.5) + 1)))
    mask = np.zeros((H, H, H), bool)
    a = np.linspace(0, 1, n)
    for i, u in enumerate(a):
        for w in a[: n - i]:
            if u + w > 1 + 1e-9:
                continue
            pts = tri[:, 0] * (1 - u - w) + tri[:, 1] * u + tri[:, 2] * w
            q = np.clip(((pts - voxel_min) / (voxel_max - voxel_min) * H).astype(np.int64), 0, H - 1)
            mask[q[:, 2], q[:, 1], q[:, 0]] = True
    kk, jj, ii = np.where(mask)
    inds = -np.ones((H, H, H), np.int64)
    inds[kk, jj, ii] = np.arange(len(ii))
    centres = (np.stack([ii, jj, kk], -1) + 0.5) / H * (voxel_max - voxel_min) + voxel_min
    return {"mask": mask, "inds": inds, "radiance": smooth_radiance(centres), "voxel_min": float(voxel_min), "voxel_max": float(voxel_max)}


def emitters_for(vertices, faces, is_emitter, radiance=(10.0, 9.0, 8.0)):
    """emitter.pth contents (extract_emitter_ldr.py:96-115): radiance is (n_face,3), rows indexed by emitter ordinal."""
    ev = vertices[faces[is_emitter]]
    cr = np.cross(ev[:, 1] - ev[:, 0], ev[:, 2] - ev[:, 0])
    area = (np.linalg.norm(cr, axis=-1) / 2.0).astype(np.float32)
    rad = np.zeros((len(faces), 3), np.float32)
    rad[: int(is_emitter.sum())] = np.asarray(radiance, np.float32)
    return {"is_emitter": is_emitter.astype(bool), "emitter_vertices": ev.astype(np.float32), "emitter_area": area, "emitter_radiance": rad}


def camera(H, W, view=0, n_views=32, eye=(2.0, 1.5, 1.3)):
    """K (3x3) with f = 0.8 W and c2w (3x4, OpenCV: x right, y down, z forward) looking around a horizontal circle."""
    K = np.array([[0.8 * W, 0, W / 2.0], [0, 0.8 * W, H / 2.0], [0, 0, 1]], np.float32)
    ang = 2 * np.pi * (view + 0.37) / n_views
    fwd = np.array([np.cos(ang), np.sin(ang), -0.12]); fwd /= np.linalg.norm(fwd)
    right = np.cross(fwd, [0, 0, 1.0]); right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    c2w = np.concatenate([np.stack([right, down, fwd], 1), np.asarray(eye, np.float64)[:, None]], 1).astype(np.float32)
    return K, c2w
