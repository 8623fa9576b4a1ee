"""Scene.World -- the reference's static world as run-time data.

`main_scene()` and `initial_camera()` restate src/Scene/World.hs:8-77 value for value (they are
data, not code).  `scene16()` is the build-defined "~16 primitives" benchmark scene of
SURVEY.md 8(d): main scene + a 3x3 grid of unit spheres.
"""
import numpy as np

MATTE, GLOSSY = 0, 1          # data Brdf = Matte Float | Glossy Float  (src/Scene/Objects.hs:77-87)
GLASS = 2                     # build-defined extension (no reference semantics); parameter = index of refraction
STREAMS, INLINE = 0, 1        # data Algorithm = Streams | Inline        (src/Scene/Trace.hs:68)

# field order = src/Scene/Objects.hs (Sphere :126-131, Plane :103-108, Material :90-100, Camera :67-74)
SPHERE_DTYPE = np.dtype([("position", "<f4", 3), ("radius", "<f4"), ("color", "<f4", 3),
                         ("illuminance", "<f4"), ("brdf_tag", "<i4"), ("brdf_param", "<f4")])
PLANE_DTYPE = np.dtype([("position", "<f4", 3), ("direction", "<f4", 3), ("color", "<f4", 3),
                        ("illuminance", "<f4"), ("brdf_tag", "<i4"), ("brdf_param", "<f4")])
CAMERA_DTYPE = np.dtype([("position", "<f4", 3), ("rotation", "<f4", 3), ("fov", "<i8")])
assert SPHERE_DTYPE.itemsize == 40 and PLANE_DTYPE.itemsize == 48 and CAMERA_DTYPE.itemsize == 32


def sphere(position, radius, color, illuminance, brdf_tag, brdf_param):
    return (tuple(position), radius, tuple(color), illuminance, brdf_tag, brdf_param)


def plane(position, direction, color, illuminance, brdf_tag, brdf_param):
    return (tuple(position), tuple(direction), tuple(color), illuminance, brdf_tag, brdf_param)


def initial_camera():
    """src/Scene/World.hs:8-12"""
    cam = np.zeros((), dtype=CAMERA_DTYPE)
    cam["position"] = (1.0, -1.6, -4.8)
    cam["rotation"] = (0.314, -0.314, 0.0)
    cam["fov"] = 90
    return cam


def camera(position, rotation, fov):
    cam = np.zeros((), dtype=CAMERA_DTYPE)
    cam["position"] = position
    cam["rotation"] = rotation
    cam["fov"] = fov
    return cam


def main_scene():
    """src/Scene/World.hs:15-77 -> (spheres, planes)"""
    spheres = np.array([
        sphere((2.0, 2.0, -14.0), 5.0, (1.0, 0.3, 0.3), 0.0, MATTE, 0.8),
        sphere((6.0, 2.0, -9.0), 1.5, (0.0, 0.4, 0.0), 0.0, MATTE, 0.9),
        sphere((4.5, 1.0, -9.0), 0.5, (0.4, 0.4, 1.0), 0.0, GLOSSY, 1.0),
        sphere((16.0, -2.05, -20.0), 0.9, (0.8, 0.8, 0.8), 6942.0, GLOSSY, 0.5),
        sphere((5.0, 10.0, 4.0), 2.0, (0.99, 0.84, 0.12), 4420.0, MATTE, 1.0),
    ], dtype=SPHERE_DTYPE)
    planes = np.array([
        plane((0.0, -3.0, 0.0), (0.0, 1.0, 0.0), (0.43, 0.95, 0.5), 0.0, MATTE, 1.5),
        plane((0.0, 15.0, 0.0), (0.0, -1.0, 0.0), (0.26, 0.68, 0.88), 0.0, GLOSSY, 0.9),
    ], dtype=PLANE_DTYPE)
    return spheres, planes


def scene16():
    """SURVEY.md 8(d) scene S16: main scene + 9 unit spheres, centres (-6+6i, 0, -6-6j)."""
    spheres, planes = main_scene()
    extra = []
    for i in range(3):
        for j in range(3):
            tag, p = (MATTE, 0.9) if (i + j) % 2 == 0 else (GLOSSY, 0.8)
            extra.append(sphere((-6.0 + 6.0 * i, 0.0, -6.0 - 6.0 * j), 1.0,
                                (0.2 + 0.3 * i, 0.5, 0.2 + 0.3 * j), 0.0, tag, p))
    spheres = np.concatenate([spheres, np.array(extra, dtype=SPHERE_DTYPE)])
    return spheres, planes


def glass_scene():
    """BASELINE.json configs[4] ("glass / refraction-heavy scene"): scene16 with the big sphere and four of the
    grid spheres turned into GLASS (ior 1.5 / 1.33).  Build-defined; the reference has no such material."""
    spheres, planes = scene16()
    spheres = spheres.copy()
    for i, ior in ((0, 1.5), (5, 1.5), (7, 1.33), (9, 1.5), (11, 1.33)):
        spheres["brdf_tag"][i] = GLASS
        spheres["brdf_param"][i] = ior
        spheres["color"][i] = (0.95, 0.95, 0.95)
    return spheres, planes


def mirror_box():
    """Test scene for LONG ray lineages (build-defined): a closed box of six inward-facing perfect mirrors
    (Glossy 1.0: the rotation angles are (1 - p) * rv = 0, so `next` is the exact reflection and no ray leaves) whose
    colour 6.2 makes a bounce keep 6.2 / (2 pi) = 98.7 % of the throughput -- a lineage takes several hundred
    traceSteps before nearZero ends it -- around a small emissive sphere.  The camera of initial_camera() is inside."""
    spheres = np.array([sphere((1.0, -1.0, -8.0), 0.6, (1.0, 0.9, 0.8), 10.0, MATTE, 1.0)], dtype=SPHERE_DTYPE)
    c, g = (6.2, 6.2, 6.2), GLOSSY
    planes = np.array([
        plane((0.0, -4.0, 0.0), (0.0, 1.0, 0.0), c, 0.0, g, 1.0), plane((0.0, 4.0, 0.0), (0.0, -1.0, 0.0), c, 0.0, g, 1.0),
        plane((-6.0, 0.0, 0.0), (1.0, 0.0, 0.0), c, 0.0, g, 1.0), plane((8.0, 0.0, 0.0), (-1.0, 0.0, 0.0), c, 0.0, g, 1.0),
        plane((0.0, 0.0, -14.0), (0.0, 0.0, 1.0), c, 0.0, g, 1.0), plane((0.0, 0.0, 2.0), (0.0, 0.0, -1.0), c, 0.0, g, 1.0),
    ], dtype=PLANE_DTYPE)
    return spheres, planes


def screen_pixels(width, height):
    """screenPixels (src/Util.hs:209-210): Matrix (V2 Int), V2 x y at index (Z :. y :. x)."""
    ys, xs = np.meshgrid(np.arange(height, dtype=np.int64), np.arange(width, dtype=np.int64), indexing="ij")
    return np.ascontiguousarray(xs), np.ascontiguousarray(ys)
