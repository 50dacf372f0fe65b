#!/usr/bin/env python3
"""Independent anchors for yaw / roll / Transform (round-3 verdict, next-round item 1b).

Writes tests/golden/zaphod_anchors.json.  NOT reference output and NOT produced by the oracle or the host
mirror: this script follows the reference's source text with closed-form trigonometry in 50-digit arithmetic
(mpmath), shares no code with oracle/ or chess2rt_amd/, reads no file of theirs, and is the only thing the
anchor test trusts.  It does for data/zaphod.sdl (the one shipped scene with yaw and roll) what the surveyor's
hand derivation did for lecture4.sdl (pitch only; SURVEY.md 8(c), tests/golden/lecture4_anchors.json).

Conventions, written out
------------------------
gfm:math 7.0.8 is an un-vendored dependency (dub.selections.json), so its conventions are restated from its
published source (`Matrix.rotateAxis!(i, j)`: identity with c[i][i] = c[j][j] = cos, c[i][j] = -sin,
c[j][i] = +sin; rotateX = rotateAxis!(1, 2), rotateY = rotateAxis!(2, 0), rotateZ = rotateAxis!(0, 1);
c[row][col], row-major; `radians(d)` = d * pi / 180):

    Rx(a) = | 1   0    0  |   Ry(a) = |  cos 0 sin |   Rz(a) = | cos -sin 0 |
            | 0  cos -sin |           |   0  1  0  |           | sin  cos 0 |
            | 0  sin  cos |           | -sin 0 cos |           |  0    0  1 |

and the reference's own `mul(v, M)` is ROW vector times matrix (rt/imported_types.d:13-20):
    (v * M)[k] = v.x * M[0][k] + v.y * M[1][k] + v.z * M[2][k].

Camera.beginFrame (rt/camera.d:101-112): rotation = Rz(roll) * Rx(pitch) * Ry(yaw), each corner / axis v
becomes v * rotation = ((v * Rz) * Rx) * Ry.  Carried out by hand for v = (a, b, c), with cr = cos(roll) etc.:

    v * Rz = (a cr + b sr,  -a sr + b cr,  c)                 =: (a1, b1, c1)
    .. * Rx = (a1,  b1 cp + c1 sp,  -b1 sp + c1 cp)            =: (a2, b2, c2)
    .. * Ry = (a2 cy - c2 sy,  b2,  a2 sy + c2 cy)

so   frontDir = (0,0,1) * R = (-cp sy,  sp,  cp cy)
     rightDir = (1,0,0) * R = ( cr cy - sr sp sy,  -sr cp,  cr sy + sr sp cy)
     upDir    = (0,1,0) * R = ( sr cy + cr sp sy,   cr cp,  sr sy - cr sp cy)

Evidence ON DISK for the three signs (the only part of gfm this derivation takes on trust), from the reference's
key bindings (gui/raytracer_demo.d:276-303, 317; Camera.rotate(dYaw, dRoll, dPitch), rt/camera.d:208-229):
  * LEFT+LSHIFT adds +4 to yaw, RIGHT+LSHIFT -4, and the mouse adds -mouseDx: positive yaw must TURN LEFT.
    frontDir.x = -cos(pitch) sin(yaw) < 0 for small positive yaw, with rightDir ~ +x: left.  (The opposite sign of
    sin in Ry would make the left key turn right.)
  * UP+LSHIFT adds +4 to pitch and the mouse adds -mouseDy: positive pitch must LOOK UP.  frontDir.y = sin(pitch).
    (Also SURVEY.md 8(c): pitch -30 in lecture4.sdl looks down at the floor.)
  * RIGHT+LCTRL adds +4 to roll: the camera must lean to the RIGHT.  rightDir.y = -sin(roll) cos(pitch) < 0: the
    right-hand side of the camera dips.
zaphod.sdl itself agrees: eye at (1.5, 17, -19.5), pitch -41.8: the centre ray meets the page (y = 0) at distance
17 / sin(41.8 deg) = 25.5, and the file asks for focalPlaneDist 25.29 — the page is in focus at the frame centre.

Transform (rt/transform.d:32-50): scale: transform = transform * diag(x, y, z); rotate: transform = transform *
Rx(pitch) * Ry(yaw) * Rz(roll); inverse and inverse-transpose follow.  Node "book" is `scale 10 10 10`:
transform = 10 I, inverse = 0.1 I.  Node.intersect (rt/node.d:24-47) takes the ray to object space (orig / 10,
dir / 10 re-normalised), Plane.intersect (rt/geometry.d:30-59) gives u = p.x, v = p.z THERE (world / 10), the point
returns through transform.point, the distance through / |dir / 10|.

Shading of a page pixel (rt/shader.d:67-105, rt/texture.d:117-128, rt/bitmap.d:48-63,116-126, rt/light.d:52-66,
rt/color.d:60-66): texel bytes through the 8-bpp palette (imageio/bmp.d:168-187; rows bottom-up, y = 0 is the top
row), byte / 255, sRGB decompression, bilinear blend with wrap-around, times lightColor * power / |p - L|^2 * cos.
Here in 50 digits; the reference does the colour part in fp32, hence `rgb_tolerance`.

Usage: python tests/golden/make_zaphod_anchors.py   (needs mpmath; the committed JSON is what the tests read)
"""
import json
import os
import struct

from mpmath import mp, mpf, cos, sin, tan, sqrt, floor, pi

mp.dps = 50
HERE = os.path.dirname(os.path.abspath(__file__))

# ---- data/zaphod.sdl, read by eye (rt/camera.d:238-254 names the fields) -------------------------------------
POS = (mpf("1.5"), mpf(17), mpf("-19.5"))
YAW, PITCH, ROLL, FOV = mpf("5.2"), mpf("-41.8"), mpf("2.3"), mpf(38)
W, H = 640, 480                      # the anchor frame (the file's own 645x430 is not needed: DOF off, AA off)
LIGHT_POS = (mpf(200), mpf(200), mpf(-200))
LIGHT_COLOR = (mpf("0.351"), mpf("0.332"), mpf("0.187"))
LIGHT_POWER = mpf(100000)
SCALE = mpf(10)
PLANE_Y = mpf(0)
PIXELS = [(320, 240), (0, 0), (639, 479), (100, 400), (500, 77), (17, 333)]


def rad(d):
    return d * pi / 180


def rot_row(v, yaw, pitch, roll):
    """v * (Rz(roll) * Rx(pitch) * Ry(yaw)), row vector, the closed form of the docstring."""
    a, b, c = v
    cr, sr, cp, sp, cy, sy = cos(rad(roll)), sin(rad(roll)), cos(rad(pitch)), sin(rad(pitch)), cos(rad(yaw)), sin(rad(yaw))
    a1, b1, c1 = a * cr + b * sr, -a * sr + b * cr, c
    a2, b2, c2 = a1, b1 * cp + c1 * sp, -b1 * sp + c1 * cp
    return (a2 * cy - c2 * sy, b2, a2 * sy + c2 * cy)


def add(a, b):
    return tuple(x + y for x, y in zip(a, b))


def sub(a, b):
    return tuple(x - y for x, y in zip(a, b))


def scl(a, s):
    return tuple(x * s for x in a)


def dot(a, b):
    return sum(x * y for x, y in zip(a, b))


def unit(a):
    return scl(a, 1 / sqrt(dot(a, a)))


def fl(v):
    return [float(x) for x in v] if isinstance(v, (tuple, list)) else float(v)


# ---- Camera.beginFrame, rt/camera.d:77-117 --------------------------------------------------------------------
aspect = mpf(W) / H
len_xy = sqrt(aspect * aspect + 1)                      # |(-aspect, 1, 1) - (0, 0, 1)|
scaling = tan(rad(FOV / 2)) / len_xy
x, y = -aspect * scaling, scaling
up_left = add(rot_row((x, y, 1), YAW, PITCH, ROLL), POS)
up_right = add(rot_row((-x, y, 1), YAW, PITCH, ROLL), POS)
down_left = add(rot_row((x, -y, 1), YAW, PITCH, ROLL), POS)
right_dir = rot_row((1, 0, 0), YAW, PITCH, ROLL)
up_dir = rot_row((0, 1, 0), YAW, PITCH, ROLL)
front_dir = rot_row((0, 0, 1), YAW, PITCH, ROLL)
# the sign statements of the docstring, checked rather than asserted in prose
assert front_dir[0] < 0 < front_dir[2] and front_dir[1] < 0 and right_dir[0] > 0 and right_dir[1] < 0 < up_dir[1]

# ---- texture: data/texture/zaphod.bmp through imageio/bmp.d:60-193 (8-bpp, palette, bottom-up) ----------------
bmp = open(os.path.join(HERE, "scenes", "texture", "zaphod.bmp"), "rb").read()
assert bmp[:2] == b"BM"
pix_off = struct.unpack_from("<I", bmp, 10)[0]
hdr_size, bw, bh, planes, bpp, compression = struct.unpack_from("<IiiHHI", bmp, 14)
colors_used = struct.unpack_from("<I", bmp, 14 + 32)[0]
assert (hdr_size, bpp, compression) == (40, 8, 0) and bh > 0
palette = [struct.unpack_from("<BBBB", bmp, 14 + hdr_size + 4 * i) for i in range(colors_used or 256)]   # B, G, R, 0
# imageio/bmp.d:172: the 8-bpp branch reads header.width bytes per row and does not skip padding
# (768 is a multiple of 4, so there is none here)
assert bw % 4 == 0


def srgb(v):                                             # rt/bitmap.d:116-126
    if v == 0:
        return mpf(0)
    if v == 1:
        return mpf(1)
    f32 = lambda s: mpf(struct.unpack("f", struct.pack("f", float(s)))[0])        # the float literals as fp32
    return v / f32("12.92") if v <= f32("0.04045") else ((v + f32("0.055")) / f32("1.055")) ** f32("2.4")


def texel(tx, ty):
    """Linear RGB of texel (tx, ty), y = 0 at the top: file row bh-1-ty (foreach_reverse, imageio/bmp.d:170)."""
    idx = bmp[pix_off + (bh - 1 - ty) * bw + tx]
    b, g, r, _ = palette[idx]
    return tuple(srgb(mpf(c) / 255) for c in (r, g, b))


def bitmap_color(u, v):                                  # rt/texture.d:117-128 (scaling 1), rt/bitmap.d:48-63
    u, v = u - floor(u), v - floor(v)
    fx, fy = u * bw, v * bh
    tx, ty = int(floor(fx)), int(floor(fy))
    p, q = fx - tx, fy - ty
    txn, tyn = (tx + 1) % bw, (ty + 1) % bh
    c00, c10, c01, c11 = texel(tx, ty), texel(txn, ty), texel(tx, tyn), texel(txn, tyn)
    rgb = tuple(c00[k] * ((1 - p) * (1 - q)) + c10[k] * (p * (1 - q)) + c01[k] * ((1 - p) * q) + c11[k] * (p * q) for k in range(3))
    return rgb, (tx, ty), (p, q)


# ---- per pixel: getScreenRay (rt/camera.d:123-147), Node.intersect, Plane.intersect, Lambert.shade ------------
pixels = []
for (px, py) in PIXELS:
    target = add(add(up_left, scl(sub(up_right, up_left), mpf(px) / W)), scl(sub(down_left, up_left), mpf(py) / H))
    d = unit(sub(target, POS))
    # object space: orig / 10, dir / 10 (length 0.1), renormalised -> same direction; plane y = 0
    o_obj = scl(POS, 1 / SCALE)
    assert o_obj[1] > PLANE_Y and d[1] < mpf("-1e-9")    # rt/geometry.d:33: the ray is not rejected
    mult_obj = (o_obj[1] - PLANE_Y) / -d[1]              # distance in object space
    p_obj = add(o_obj, scl(d, mult_obj))
    u, v = p_obj[0], p_obj[2]                            # rt/geometry.d:52-53
    p_world = scl(p_obj, SCALE)                          # transform.point
    dist_world = mult_obj / (1 / SCALE)                  # data.dist /= rayDirLength, rayDirLength = |dir / 10| = 0.1
    n = (mpf(0), mpf(1), mpf(0))                         # normalize((0,1,0) * 0.1 I); faceforward keeps it: d.y < 0
    diffuse, (tx, ty), (fp, fq) = bitmap_color(u, v)
    to_light = sub(LIGHT_POS, p_world)
    dist2 = dot(to_light, to_light)
    cos_theta = dot(unit(to_light), n)
    assert cos_theta > 0                                 # and nothing can shadow the page: it is the only node,
    #                                                      the shadow ray starts above it and points up (visible)
    rgb = tuple(diffuse[k] * (LIGHT_COLOR[k] * LIGHT_POWER / dist2 * cos_theta) for k in range(3))
    pixels.append({"x": px, "y": py, "dir": fl(d), "t": fl(dist_world), "p": fl(p_world), "u": fl(u), "v": fl(v),
                   "texel": [tx, ty], "bilinear_pq": [fl(fp), fl(fq)], "rgb": fl(rgb)})

# ---- Transform (rt/transform.d): the node of zaphod.sdl, and the real rotate the loader never reaches ----------
def matmul(a, b):
    return [[sum(a[i][k] * b[k][j] for k in range(3)) for j in range(3)] for i in range(3)]


def Rx(a):
    return [[1, 0, 0], [0, cos(a), -sin(a)], [0, sin(a), cos(a)]]


def Ry(a):
    return [[cos(a), 0, sin(a)], [0, 1, 0], [-sin(a), 0, cos(a)]]


def Rz(a):
    return [[cos(a), -sin(a), 0], [sin(a), cos(a), 0], [0, 0, 1]]


def transpose(m):
    return [[m[j][i] for j in range(3)] for i in range(3)]


T_YAW, T_PITCH, T_ROLL = mpf(30), mpf(20), mpf(10)
rot = matmul(matmul(Rx(rad(T_PITCH)), Ry(rad(T_YAW))), Rz(rad(T_ROLL)))        # identity * Rx * Ry * Rz
rot_inv = transpose(rot)                                                       # orthonormal: inverse = transpose
sc = (mpf(2), mpf(3), mpf(5))
scaled_rot = [[sc[i] * rot[i][j] for j in range(3)] for i in range(3)]         # diag(s) * R  (scale first, then rotate)
scaled_rot_inv = [[rot_inv[i][j] / sc[j] for j in range(3)] for i in range(3)]  # R^T * diag(1/s)
transform_cases = [
    {"ops": [["scale", 10, 10, 10]], "transform": [[10, 0, 0], [0, 10, 0], [0, 0, 10]],
     "inverse": [[0.1, 0, 0], [0, 0.1, 0], [0, 0, 0.1]], "transposed_inverse": [[0.1, 0, 0], [0, 0.1, 0], [0, 0, 0.1]],
     "offset": [0, 0, 0], "note": "node 'book' of zaphod.sdl"},
    {"ops": [["rotate", fl(T_YAW), fl(T_PITCH), fl(T_ROLL)]], "transform": [fl(r) for r in rot],
     "inverse": [fl(r) for r in rot_inv], "transposed_inverse": [fl(r) for r in rot], "offset": [0, 0, 0],
     "note": "Transform.rotate(yaw, pitch, roll) = Rx(pitch) * Ry(yaw) * Rz(roll), rt/transform.d:41-50"},
    {"ops": [["scale", 2, 3, 5], ["rotate", fl(T_YAW), fl(T_PITCH), fl(T_ROLL)], ["translate", -1, 2, -3]],
     "transform": [fl(r) for r in scaled_rot], "inverse": [fl(r) for r in scaled_rot_inv],
     "transposed_inverse": [fl(r) for r in transpose(scaled_rot_inv)], "offset": [-1, 2, -3],
     "note": "diag(2,3,5) * Rx * Ry * Rz; translate only sets the offset (rt/transform.d:52-55)"},
]
# a point through the third transform: point(P) = P * transform + offset (rt/transform.d:57-63)
P = (mpf("0.5"), mpf("-1.25"), mpf(2))
pt = tuple(sum(P[i] * scaled_rot[i][k] for i in range(3)) for k in range(3))
transform_cases[2]["point_in"] = fl(P)
transform_cases[2]["point_out"] = fl(add(pt, (mpf(-1), mpf(2), mpf(-3))))

out = {
    "source": "tests/golden/make_zaphod_anchors.py: closed-form derivation in 50-digit arithmetic following rt/camera.d:77-147, "
              "rt/transform.d:32-63, rt/node.d:24-47, rt/geometry.d:30-59, rt/texture.d:117-128, rt/bitmap.d:48-63,116-126, "
              "rt/shader.d:67-105 with the gfm conventions written out in the script's docstring; NOT reference output, "
              "NOT oracle / host-mirror output",
    "scene": "zaphod.sdl", "width": W, "height": H, "taps": 1, "dof": 0,
    "camera": {"up_left": fl(up_left), "up_right": fl(up_right), "down_left": fl(down_left), "right_dir": fl(right_dir),
               "up_dir": fl(up_dir), "front_dir": fl(front_dir), "tolerance": 1e-13},
    "geom_tolerance": 1e-11, "dir_tolerance": 1e-14, "rgb_tolerance": 2e-6, "matrix_tolerance": 1e-15,
    "pixels": pixels, "transforms": transform_cases,
}
with open(os.path.join(HERE, "zaphod_anchors.json"), "w") as f:
    json.dump(out, f, indent=1)
print("wrote zaphod_anchors.json:", len(pixels), "pixels,", len(transform_cases), "transforms")
for p in pixels:
    print(p["x"], p["y"], p["t"], p["texel"], p["rgb"])
