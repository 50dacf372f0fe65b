// scene.cpp — host-side mirror of the reference's scene build: loader
// (rt/scene_loader.d + the deserialize() methods), Camera, Transform, BMP
// decode and the Scene -> flat-table converter.  Cold code: runs once per
// scene / frame on the CPU, exactly where the reference runs it.
#include "scene.hpp"

#include <algorithm>
#include <cstring>
#include <fstream>
#include <mutex>
#include <sstream>

namespace c2rt {
namespace host {

// ------------------------------------------------------------------ gfm:math
Matrix Matrix::identity()
{
    Matrix m;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) m.c[i][j] = i == j ? 1.0 : 0.0;
    return m;
}
static Matrix rotateAxis(int i, int j, double angle)
{
    Matrix r = Matrix::identity();
    const double cosa = std::cos(angle), sina = std::sin(angle);
    r.c[i][i] = cosa;
    r.c[i][j] = -sina;
    r.c[j][i] = sina;
    r.c[j][j] = cosa;
    return r;
}
Matrix Matrix::rotateX(double a) { return rotateAxis(1, 2, a); }
Matrix Matrix::rotateY(double a) { return rotateAxis(2, 0, a); }
Matrix Matrix::rotateZ(double a) { return rotateAxis(0, 1, a); }
Matrix Matrix::operator*(const Matrix &b) const
{
    Matrix r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double sum = 0;
            for (int k = 0; k < 3; ++k) sum += c[i][k] * b.c[k][j];
            r.c[i][j] = sum;
        }
    return r;
}
Matrix Matrix::inverse() const
{
    const double det = c[0][0] * (c[1][1] * c[2][2] - c[2][1] * c[1][2]) -
                       c[0][1] * (c[1][0] * c[2][2] - c[1][2] * c[2][0]) +
                       c[0][2] * (c[1][0] * c[2][1] - c[1][1] * c[2][0]);
    const double invDet = 1 / det;
    Matrix r;
    r.c[0][0] = (c[1][1] * c[2][2] - c[2][1] * c[1][2]) * invDet;
    r.c[0][1] = -(c[0][1] * c[2][2] - c[0][2] * c[2][1]) * invDet;
    r.c[0][2] = (c[0][1] * c[1][2] - c[0][2] * c[1][1]) * invDet;
    r.c[1][0] = -(c[1][0] * c[2][2] - c[1][2] * c[2][0]) * invDet;
    r.c[1][1] = (c[0][0] * c[2][2] - c[0][2] * c[2][0]) * invDet;
    r.c[1][2] = -(c[0][0] * c[1][2] - c[1][0] * c[0][2]) * invDet;
    r.c[2][0] = (c[1][0] * c[2][1] - c[2][0] * c[1][1]) * invDet;
    r.c[2][1] = -(c[0][0] * c[2][1] - c[2][0] * c[0][1]) * invDet;
    r.c[2][2] = (c[0][0] * c[1][1] - c[1][0] * c[0][1]) * invDet;
    return r;
}
Matrix Matrix::transposed() const
{
    Matrix r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r.c[i][j] = c[j][i];
    return r;
}
Vector mul(const Vector &v, const Matrix &m)
{
    return Vector(v.x * m.c[0][0] + v.y * m.c[1][0] + v.z * m.c[2][0],
                  v.x * m.c[0][1] + v.y * m.c[1][1] + v.z * m.c[2][1],
                  v.x * m.c[0][2] + v.y * m.c[1][2] + v.z * m.c[2][2]);
}
/* gfm radians!double: `return x * (PI / 180);` — std.math.PI is an 80-bit `real`, so the constant is
 * folded in extended precision, x is promoted, the product is formed on the x87 (64-bit significand)
 * and rounded to double once on return.  x86-64 `long double` is that format. */
double radians(double deg)
{
    static_assert(sizeof(long double) >= 10 && __LDBL_MANT_DIG__ == 64, "x87 extended precision expected on the host");
    return (double)((long double)deg * (3.141592653589793238462643383279502884L / 180.0L));
}

static Vector operator+(const Vector &a, const Vector &b) { return Vector(a.x + b.x, a.y + b.y, a.z + b.z); }
static Vector operator-(const Vector &a, const Vector &b) { return Vector(a.x - b.x, a.y - b.y, a.z - b.z); }
static Vector operator*(double s, const Vector &a) { return Vector(a.x * s, a.y * s, a.z * s); }
static double magnitude(const Vector &a)
{
    double s = 0;
    s += a.x * a.x;
    s += a.y * a.y;
    s += a.z * a.z;
    return std::sqrt(s);
}

// ------------------------------------------------------------------ Camera
void Camera::beginFrame() // rt/camera.d:77-117
{
    double x = -aspect;
    double y = +1;
    const Vector corner(x, y, 1), center(0, 0, 1);
    const double lenXY = magnitude(corner - center);
    const double wantedLength = std::tan(radians(fov / 2));
    const double scaling = wantedLength / lenXY;
    x *= scaling;
    y *= scaling;
    upLeft = Vector(x, y, 1);
    upRight = Vector(-x, y, 1);
    downLeft = Vector(x, -y, 1);
    const Matrix rotation = Matrix::rotateZ(radians(roll)) * Matrix::rotateX(radians(pitch)) * Matrix::rotateY(radians(yaw));
    upLeft = mul(upLeft, rotation);
    upRight = mul(upRight, rotation);
    downLeft = mul(downLeft, rotation);
    rightDir = mul(Vector(1, 0, 0), rotation);
    upDir = mul(Vector(0, 1, 0), rotation);
    frontDir = mul(Vector(0, 0, 1), rotation);
    upLeft = upLeft + pos;
    upRight = upRight + pos;
    downLeft = downLeft + pos;
}
void Camera::move(double dx, double dy, double dz) // :181-206
{
    pos = pos + dx * rightDir;
    pos = pos + dy * upDir;
    pos = pos + dz * frontDir;
}
void Camera::rotate(double dYaw, double dRoll, double dPitch) // :212-229
{
    yaw += dYaw;
    roll += dRoll;
    pitch += dPitch;
    pitch = std::min(std::max(pitch, -90.0), 90.0);
}
void Camera::setFrameSize(uint32_t w, uint32_t h) // :231-236
{
    frameWidth = w;
    frameHeight = h;
    aspect = double(frameWidth) / double(frameHeight);
}
void Camera::fill(c2rt_camera_frame &o) const
{
    std::memset(&o, 0, sizeof o);
    const Vector *src[7] = {&pos, &upLeft, &upRight, &downLeft, &rightDir, &upDir, &frontDir};
    double *dst[7] = {o.pos, o.up_left, o.up_right, o.down_left, o.right_dir, o.up_dir, o.front_dir};
    for (int i = 0; i < 7; ++i) { dst[i][0] = src[i]->x; dst[i][1] = src[i]->y; dst[i][2] = src[i]->z; }
    o.frame_width = (double)frameWidth;
    o.frame_height = (double)frameHeight;
    o.dof = dof ? 1u : 0u;
    o.num_samples = (uint32_t)numSamples;
    o.focal_plane_dist = focalPlaneDist;
    o.disc_multiplier = discMultiplier;
    o.stereo_separation = stereoSeparation;
}

// --------------------------------------------------------------- Transform
void Transform::reset() // rt/transform.d:24-30
{
    transform = Matrix::identity();
    inverseTransform = transform.inverse();
    transposedInverse = inverseTransform.transposed();
    offset = Vector(0.0, 0.0, 0.0);
}
void Transform::scale(double x, double y, double z) // :32-39, scaledIdentity rt/imported_types.d:22-29
{
    Matrix scaling;
    std::memset(&scaling, 0, sizeof scaling);
    scaling.c[0][0] = x;
    scaling.c[1][1] = y;
    scaling.c[2][2] = z;
    transform = transform * scaling;
    inverseTransform = transform.inverse();
    transposedInverse = inverseTransform.transposed();
}
void Transform::rotate(double yaw, double pitch, double roll) // :41-50
{
    transform = transform * Matrix::rotateX(radians(pitch)) * Matrix::rotateY(radians(yaw)) * Matrix::rotateZ(radians(roll));
    inverseTransform = transform.inverse();
    transposedInverse = inverseTransform.transposed();
}
void Transform::translate(const Vector &v) { offset = v; } // :52-55

// ------------------------------------------------------------------ colour
namespace {
uint8_t g_srgb_cache[4097];
std::once_flag g_srgb_once;
uint8_t roundToByte(float x) { return (uint8_t)(int)std::floor(x * 255.0f); } // rt/color.d:216-219
uint8_t convertTo8bit_sRGB(float x) // rt/color.d:194-207
{
    if (x <= 0) return 0;
    if (x >= 1) return 255;
    if (x <= 0.0031308f) x = x * 12.02f;
    else x = (float)(1.055 * std::pow((double)x, 1 / 2.4) - 0.055);
    return roundToByte(x);
}
} // namespace
uint32_t colorToRGB32(const float rgb[3])
{
    std::call_once(g_srgb_once, [] { for (int i = 0; i < 4097; ++i) g_srgb_cache[i] = convertTo8bit_sRGB(i / 4096.0f); });
    uint32_t ch[3];
    for (int i = 0; i < 3; ++i) {
        const float x = rgb[i];
        ch[i] = !(x > 0) ? 0u : (x >= 1 ? 255u : (uint32_t)g_srgb_cache[(int)(x * 4096.0f)]); // :209-214
    }
    return ch[2] | (ch[1] << 8) | (ch[0] << 16); // toRGB32, :154-162
}

// --------------------------------------------------------------------- BMP
namespace {
uint32_t rd32(const uint8_t *p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
uint32_t rd16(const uint8_t *p) { return (uint32_t)(p[0] | (p[1] << 8)); }
[[noreturn]] void bmp_fail(const std::string &m) { throw SceneError(C2RT_ERR_PARSE, "BMP: " + m); }
} // namespace

Bitmap loadBmpImage(const uint8_t *b, size_t len) // imageio/bmp.d:41-193
{
    if (!b || len < 14 + 12) bmp_fail("file too short");
    if (b[0] != 'B' || b[1] != 'M') bmp_fail("Only files beginning with 'BM' are supported!");
    const uint32_t pixelOffset = rd32(b + 10);
    const uint32_t ver = rd32(b + 14);
    int64_t w, h;
    uint32_t planes, bpp, colorsUsed = 0;
    size_t palElem;
    if (ver == 12) {
        w = (int16_t)rd16(b + 18); h = (int16_t)rd16(b + 20); planes = rd16(b + 22); bpp = rd16(b + 24); palElem = 3;
    } else if (ver == 40 || ver == 52 || ver == 56 || ver == 108 || ver == 124) {
        if (len < 14 + 40) bmp_fail("truncated header");
        w = (int32_t)rd32(b + 18); h = (int32_t)rd32(b + 22); planes = rd16(b + 26); bpp = rd16(b + 28);
        colorsUsed = rd32(b + 46); palElem = 4;
    } else
        bmp_fail("unsupported DIB header size " + std::to_string(ver));
    if (planes != 1) bmp_fail("Only .bmp files with 1 color plane are supported.");
    if (!(bpp == 1 || bpp == 2 || bpp == 4 || bpp == 8 || bpp == 16 || bpp == 24 || bpp == 32 || bpp == 64))
        bmp_fail("unsupported bpp " + std::to_string(bpp));
    if (!(bpp == 8 || bpp == 24 || bpp == 32))
        bmp_fail("bpp " + std::to_string(bpp) + " is not decodable by the reference either (imageio/bmp.d:167-190)");
    if (w <= 0 || h <= 0) bmp_fail("non-positive size");
    const uint8_t *palette = b + 14 + ver;
    uint32_t paletteSize = 0;
    if (bpp == 8) {
        paletteSize = (ver != 12 && colorsUsed) ? colorsUsed : 256u;
        if ((size_t)(14 + ver) + (size_t)paletteSize * palElem > len) bmp_fail("truncated palette");
    }
    const size_t rowBytes = (size_t)(bpp / 8) * (size_t)w;
    const size_t rowPadded = (((size_t)bpp * (size_t)w + 31) / 32) * 4;
    Bitmap bm;
    bm.width = (uint32_t)w;
    bm.height = (uint32_t)h;
    bm.pixels.resize((size_t)w * h * 3);
    const float divider = 1.0f / 255.0f; // Color(uint) — rt/color.d:60-66
    size_t cur = pixelOffset;
    for (int64_t y = h - 1; y >= 0; --y) { // file rows are bottom-up
        const size_t need = bpp == 8 ? (size_t)w : rowBytes;
        if (cur + need > len) bmp_fail("truncated pixel array");
        float *row = bm.pixels.data() + (size_t)y * w * 3;
        for (int64_t x = 0; x < w; ++x) {
            uint32_t rgb;
            if (bpp == 24) rgb = b[cur + 3 * x] | (b[cur + 3 * x + 1] << 8) | (b[cur + 3 * x + 2] << 16);
            else if (bpp == 32) rgb = rd32(b + cur + 4 * x);
            else {
                const uint32_t idx = b[cur + x];
                if (idx >= paletteSize) bmp_fail("palette index out of range");
                const uint8_t *pe = palette + (size_t)idx * palElem;
                rgb = pe[0] | (pe[1] << 8) | (pe[2] << 16);
            }
            row[3 * x + 0] = (float)((rgb >> 16) & 0xff) * divider;
            row[3 * x + 1] = (float)((rgb >> 8) & 0xff) * divider;
            row[3 * x + 2] = (float)(rgb & 0xff) * divider;
        }
        // the <=8-bpp branch reads `width` bytes per row and never skips padding (:167-170)
        cur += bpp == 8 ? (size_t)w : rowPadded;
    }
    return bm;
}

void applyAssumedGamma(float *t, size_t n, float assumedGamma) // rt/texture.d:137-141, rt/bitmap.d:116-136
{
    if (assumedGamma == 2.2f) {
        for (size_t i = 0; i < n; ++i) {
            const float x = t[i];
            if (x == 0) t[i] = 0.0f;
            else if (x == 1) t[i] = 1.0f;
            else if (x <= 0.04045f) t[i] = x / 12.92f;
            else t[i] = (float)std::pow((double)((x + 0.055f) / 1.055f), (double)2.4f);
        }
    } else if (assumedGamma != 1 && assumedGamma > 0 && assumedGamma < 10) {
        for (size_t i = 0; i < n; ++i) {
            const float x = t[i];
            if (x == 0) t[i] = 0.0f;
            else if (x == 1) t[i] = 1.0f;
            else t[i] = (float)std::pow((double)x, (double)assumedGamma);
        }
    }
}

std::vector<uint8_t> saveBmp(const float *rgb, uint32_t width, uint32_t height) // imageio/bmp.d:195-237
{
    const uint32_t fileSize = 14u + 40u + 3u * width * height;
    std::vector<uint8_t> f(fileSize, 0);
    auto w32 = [&](size_t o, uint32_t v) { f[o] = v & 0xff; f[o + 1] = (v >> 8) & 0xff; f[o + 2] = (v >> 16) & 0xff; f[o + 3] = (v >> 24) & 0xff; };
    auto w16 = [&](size_t o, uint32_t v) { f[o] = v & 0xff; f[o + 1] = (v >> 8) & 0xff; };
    f[0] = 'B'; f[1] = 'M';
    w32(2, fileSize);
    w32(10, 14 + 40);
    w32(14, 40);
    w32(18, width);
    w32(22, height);
    w16(26, 1);
    w16(28, 24);
    w32(30, 0);
    w32(34, fileSize - (14 + 40));
    const uint32_t ppm = (uint32_t)std::lrint(72 * 100.0 / 2.54);
    w32(38, ppm);
    w32(42, ppm);
    size_t o = 54;
    for (int64_t y = (int64_t)height - 1; y >= 0; --y)
        for (uint32_t x = 0; x < width; ++x) {
            const uint32_t px = colorToRGB32(rgb + ((size_t)y * width + x) * 3);
            f[o++] = px & 0xff;
            f[o++] = (px >> 8) & 0xff;
            f[o++] = (px >> 16) & 0xff;
        }
    return f;
}

// ------------------------------------------------------------------- loader
namespace {

std::string dirName(const std::string &p)
{
    const size_t s = p.find_last_of('/');
    if (s == std::string::npos) return ".";
    if (s == 0) return "/";
    return p.substr(0, s);
}
std::string toLower(std::string s)
{
    for (char &c : s) c = (char)std::tolower((unsigned char)c);
    return s;
}
std::string extension(const std::string &p)
{
    const size_t s = p.find_last_of('/');
    const size_t d = p.find_last_of('.');
    if (d == std::string::npos || (s != std::string::npos && d < s)) return "";
    return p.substr(d);
}
bool readFile(const std::string &path, std::string &out)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::ostringstream ss;
    ss << f.rdbuf();
    out = ss.str();
    return true;
}

// SceneLoadContext — rt/scene_loader.d:87-204
struct SceneLoadContext {
    Scene *scene;
    std::string baseDir;
    std::map<std::string, int> lights, geometries, textures, shaders, nodes; // NamedEntities

    // set / setTo for scalars — :99-133,144-147
    bool set(bool &p, const SceneDscNode &v, const char *name) { if (!v.isSpecified(name)) return false; p = v.getChild(name)->getBool(); return true; }
    bool set(uint32_t &p, const SceneDscNode &v, const char *name)
    {
        if (!v.isSpecified(name)) return false;
        const long long i = v.getChild(name)->getInt();
        if (i < 0 || i > 0xffffffffLL) throw SceneError(C2RT_ERR_PARSE, std::string(name) + ": value out of range"); // to!uint throws
        p = (uint32_t)i;
        return true;
    }
    bool set(uint64_t &p, const SceneDscNode &v, const char *name)
    {
        if (!v.isSpecified(name)) return false;
        const long long i = v.getChild(name)->getInt();
        if (i < 0) throw SceneError(C2RT_ERR_PARSE, std::string(name) + ": value out of range");
        p = (uint64_t)i;
        return true;
    }
    bool set(double &p, const SceneDscNode &v, const char *name) { if (!v.isSpecified(name)) return false; p = v.getChild(name)->getFloat(); return true; }
    bool set(float &p, const SceneDscNode &v, const char *name) { if (!v.isSpecified(name)) return false; p = (float)v.getChild(name)->getFloat(); return true; }
    bool set(std::string &p, const SceneDscNode &v, const char *name) { if (!v.isSpecified(name)) return false; p = v.getChild(name)->getString(); return true; }
    static double num(const DscValue &d)
    {
        if (d.kind == DscValue::Float) return d.f;
        if (d.kind == DscValue::Int) return (double)d.i;
        throw SceneError(C2RT_ERR_PARSE, "expected a number");
    }
    static void three(const SceneDscNode &c, double out[3]) // Vector | Color from 3 values — :152-157
    {
        const std::vector<DscValue> vals = c.getValues();
        if (vals.size() < 3) throw SceneError(C2RT_ERR_PARSE, "expected 3 values");
        for (int i = 0; i < 3; ++i) out[i] = num(vals[i]);
    }
    bool set(Vector &p, const SceneDscNode &v, const char *name)
    {
        if (!v.isSpecified(name)) return false;
        double d[3];
        three(*v.getChild(name), d);
        p = Vector(d[0], d[1], d[2]);
        return true;
    }
    bool set(Color &p, const SceneDscNode &v, const char *name)
    {
        if (!v.isSpecified(name)) return false;
        double d[3];
        three(*v.getChild(name), d);
        p = Color((float)d[0], (float)d[1], (float)d[2]);
        return true;
    }
    bool set(std::vector<double> &p, const SceneDscNode &v, const char *name) // scalar arrays — :169-171
    {
        if (!v.isSpecified(name)) return false;
        p.clear();
        for (const DscValue &d : v.getChild(name)->getValues()) p.push_back(num(d));
        return true;
    }
    bool set(std::vector<Color> &p, const SceneDscNode &v, const char *name) // struct arrays — :172-174
    {
        if (!v.isSpecified(name)) return false;
        p.clear();
        for (const auto &c : v.getChild(name)->getChildren()) {
            double d[3];
            three(*c, d);
            p.emplace_back((float)d[0], (float)d[1], (float)d[2]);
        }
        return true;
    }
    static int lookup(const std::map<std::string, int> &m, const std::string &key, const char *what)
    {
        const auto it = m.find(key);
        if (it == m.end()) throw SceneError(C2RT_ERR_PARSE, std::string("unknown ") + what + " '" + key + "'"); // D: RangeError
        return it->second;
    }
    static void registerName(std::map<std::string, int> &m, const SceneDscNode &v, int index, std::string &nameOut)
    {
        std::string name;
        if (v.getName(name)) {
            if (m.count(name)) throw SceneError(C2RT_ERR_PARSE, "EntityWithDuplicateName: " + name); // :196-198
            m[name] = index;
            nameOut = name;
        }
    }
};

double clampd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); } // gfm.math.clamp

void deserializeSettings(GlobalSettings &s, const SceneDscNode &v, SceneLoadContext &c) // rt/global_settings.d:47-71
{
    c.set(s.frameWidth, v, "frameWidth");
    c.set(s.frameHeight, v, "frameHeight");
    c.set(s.fullscreen, v, "fullscreen");
    c.set(s.allowResize, v, "allowResize");
    c.set(s.dynamicAspectRatio, v, "dynamicAspectRatio");
    c.set(s.interactive, v, "interactive");
    c.set(s.bucketSize, v, "bucketSize");
    c.set(s.threadCount, v, "threadCount");
    c.set(s.prepassEnabled, v, "prepassEnabled");
    c.set(s.prepassOnly, v, "prepassOnly");
    c.set(s.GIEnabled, v, "GIEnabled");
    c.set(s.AAEnabled, v, "AAEnabled");
    c.set(s.AAThreshold, v, "AAThreshold");
    c.set(s.maxTraceDepth, v, "maxTraceDepth");
    c.set(s.pathsPerPixel, v, "pathsPerPixel");
    c.set(s.ambientLightColor, v, "ambientLightColor");
    c.set(s.debugEnabled, v, "debugEnabled");
}

void deserializeCamera(Camera &cam, const SceneDscNode &v, SceneLoadContext &c) // rt/camera.d:238-255
{
    c.set(cam.pos, v, "pos");
    c.set(cam.yaw, v, "yaw");
    c.set(cam.pitch, v, "pitch");
    c.set(cam.roll, v, "roll");
    c.set(cam.fov, v, "fov");
    c.set(cam.focalPlaneDist, v, "focalPlaneDist");
    c.set(cam.fNumber, v, "fNumber");
    c.set(cam.dof, v, "dof");
    c.set(cam.numSamples, v, "numSamples");
    c.set(cam.stereoSeparation, v, "stereoSeparation");
    cam.discMultiplier = 10.0 / cam.fNumber;
    cam.setFrameSize(c.scene->settings.frameWidth, c.scene->settings.frameHeight);
}

// createObject + makeInstanceOf — rt/scene_loader.d:183-203, util/factory2.d:5-23
void requireType(const std::string &type, std::initializer_list<const char *> allowed)
{
    for (const char *a : allowed) if (type == a) return;
    throw SceneError(C2RT_ERR_PARSE, "Unknown object type (or not yet supported): " + type);
}

void loadLights(const SceneDscNode &list, SceneLoadContext &c)
{
    for (const auto &e : list.getChildren()) {
        requireType(e->getType(), {"PointLight"});
        Light l;
        c.set(l.lightColor, *e, "color"); // rt/light.d:39-43
        c.set(l.lightPower, *e, "power");
        c.set(l.pos, *e, "pos");          // :77-82
        c.scene->lights.push_back(l);
        SceneLoadContext::registerName(c.lights, *e, (int)c.scene->lights.size() - 1, c.scene->lights.back().name);
    }
}

void loadGeometries(const SceneDscNode &list, SceneLoadContext &c)
{
    for (const auto &e : list.getChildren()) {
        const std::string type = e->getType();
        requireType(type, {"Plane", "Sphere", "Cube", "CsgUnion", "CsgInter", "CsgDiff"});
        Geometry g;
        if (type == "Plane") { // rt/geometry.d:61-64
            g.type = C2RT_GEOM_PLANE;
            c.set(g.y, *e, "y");
        } else if (type == "Sphere") { // :132-140
            g.type = C2RT_GEOM_SPHERE;
            if (!c.set(g.center, *e, "center")) g.center = Vector(0, 0, 0);
            c.set(g.R, *e, "R");
        } else if (type == "Cube") { // :237-241
            g.type = C2RT_GEOM_CUBE;
            c.set(g.center, *e, "center");
            c.set(g.side, *e, "side");
        } else { // CsgOp.deserialize — :339-348
            g.type = type == "CsgUnion" ? C2RT_GEOM_CSG_UNION : (type == "CsgInter" ? C2RT_GEOM_CSG_INTER : C2RT_GEOM_CSG_DIFF);
            std::string n;
            c.set(n, *e, "left");
            g.left = SceneLoadContext::lookup(c.geometries, n, "geometry");
            n.clear();
            c.set(n, *e, "right");
            g.right = SceneLoadContext::lookup(c.geometries, n, "geometry");
        }
        c.scene->geometries.push_back(g);
        SceneLoadContext::registerName(c.geometries, *e, (int)c.scene->geometries.size() - 1, c.scene->geometries.back().name);
    }
}

void loadTextures(const SceneDscNode &list, SceneLoadContext &c)
{
    for (const auto &e : list.getChildren()) {
        const std::string type = e->getType();
        requireType(type, {"Checker", "Procedure2", "BitmapTexture"});
        Texture t;
        if (type == "Checker") { // rt/texture.d:56-61
            t.type = C2RT_TEX_CHECKER;
            c.set(t.color1, *e, "color1");
            c.set(t.color2, *e, "color2");
            c.set(t.size, *e, "size");
        } else if (type == "Procedure2") { // :88-94
            t.type = C2RT_TEX_PROCEDURE2;
            c.set(t.colorU, *e, "colorU");
            c.set(t.colorV, *e, "colorV");
            c.set(t.freqU, *e, "freqU");
            c.set(t.freqV, *e, "freqV");
            if (t.colorU.size() < 3 || t.colorV.size() < 3 || t.freqU.size() < 3 || t.freqV.size() < 3)
                throw SceneError(C2RT_ERR_PARSE, "Procedure2 needs 3 colours and 3 frequencies per axis (rt/texture.d:81-83)");
        } else { // BitmapTexture.deserialize — :128-142
            t.type = C2RT_TEX_BITMAP;
            c.set(t.scaling, *e, "scaling");
            c.set(t.assumedGamma, *e, "assumedGamma");
            std::string file;
            c.set(file, *e, "file");
            const std::string path = (!file.empty() && file[0] == '/') ? file : c.baseDir + "/" + file; // resolveRelativePath :135-138
            std::string bytes;
            if (toLower(extension(path)) != ".bmp") throw SceneError(C2RT_ERR_PARSE, "UnknownImageTypeException: " + path); // rt/bitmap.d:74-79
            if (!readFile(path, bytes)) throw SceneError(C2RT_ERR_IO, "cannot read texture file " + path);
            t.bmp = loadBmpImage(reinterpret_cast<const uint8_t *>(bytes.data()), bytes.size());
            applyAssumedGamma(t.bmp.pixels.data(), t.bmp.pixels.size(), t.assumedGamma);
        }
        c.scene->textures.push_back(std::move(t));
        SceneLoadContext::registerName(c.textures, *e, (int)c.scene->textures.size() - 1, c.scene->textures.back().name);
    }
}

void loadShaders(const SceneDscNode &list, SceneLoadContext &c)
{
    for (const auto &e : list.getChildren()) {
        const std::string type = e->getType();
        requireType(type, {"Lambert", "Phong"});
        Shader s;
        s.type = type == "Phong" ? C2RT_SHADER_PHONG : C2RT_SHADER_LAMBERT;
        c.set(s.color, *e, "color"); // Shader.deserialize — rt/shader.d:40-44
        if (s.type == C2RT_SHADER_PHONG) { // :263-280
            c.set(s.exponent, *e, "exponent");
            s.exponent = clampd(s.exponent, 1e-6, 1e6);
            c.set(s.strength, *e, "strength");
            s.strength = (float)clampd(s.strength, 0, 1e6);
        }
        std::string t;
        c.set(t, *e, "texture"); // optional — :137-146,273-279
        const auto it = c.textures.find(t);
        s.texture = it != c.textures.end() ? it->second : -1;
        c.scene->shaders.push_back(s);
        SceneLoadContext::registerName(c.shaders, *e, (int)c.scene->shaders.size() - 1, c.scene->shaders.back().name);
    }
}

void loadNodes(const SceneDscNode &list, SceneLoadContext &c)
{
    for (const auto &e : list.getChildren()) {
        requireType(e->getType(), {"Node"});
        Node n; // rt/node.d:70-94
        std::string geom, shad, bump;
        c.set(geom, *e, "geometry");
        c.set(shad, *e, "shader");
        c.set(bump, *e, "bump");
        n.geom = SceneLoadContext::lookup(c.geometries, geom, "geometry");
        n.shader = SceneLoadContext::lookup(c.shaders, shad, "shader");
        const auto it = c.textures.find(bump);
        n.bumpmap = it != c.textures.end() ? it->second : -1;
        Vector v;
        if (c.set(v, *e, "scale")) n.transform.scale(v.x, v.y, v.z);
        if (c.set(v, *e, "rotate")) n.transform.scale(v.x, v.y, v.z); // sic: rt/node.d:89-90
        if (c.set(v, *e, "translate")) n.transform.translate(v);
        c.scene->nodes.push_back(n);
        SceneLoadContext::registerName(c.nodes, *e, (int)c.scene->nodes.size() - 1, c.scene->nodes.back().name);
    }
}

// loadFromAbstractDataFormat — rt/scene_loader.d:62-83 (fixed section order)
std::unique_ptr<Scene> loadFromAbstractDataFormat(const SceneDscNode &val, const std::string &baseDir)
{
    std::unique_ptr<Scene> scene(new Scene());
    SceneLoadContext c;
    c.scene = scene.get();
    c.baseDir = baseDir;
    c.set(scene->name, val, "Name");
    if (val.isSpecified("GlobalSettings")) {
        const auto n = val.getChild("GlobalSettings");
        requireType(n->getType(), {"GlobalSettings"});
        deserializeSettings(scene->settings, *n, c);
    }
    if (val.isSpecified("Camera")) {
        const auto n = val.getChild("Camera");
        requireType(n->getType(), {"Camera"});
        deserializeCamera(scene->camera, *n, c);
    }
    if (val.isSpecified("Environment")) requireType(val.getChild("Environment")->getType(), {"Environment"});
    if (val.isSpecified("Lights")) loadLights(*val.getChild("Lights"), c);
    if (val.isSpecified("Geometries")) loadGeometries(*val.getChild("Geometries"), c);
    if (val.isSpecified("Textures")) loadTextures(*val.getChild("Textures"), c);
    if (val.isSpecified("Shaders")) loadShaders(*val.getChild("Shaders"), c);
    if (val.isSpecified("Nodes")) loadNodes(*val.getChild("Nodes"), c);
    return scene;
}

} // namespace

std::unique_ptr<Scene> parseSceneFromString(const std::string &data, const std::string &ext_, const std::string &baseDir)
{
    const std::string ext = toLower(ext_);
    if (ext == ".json") { // readAndParseData — rt/scene_loader.d:47-60
        const JsonValue root = parseJson(data);
        if (root.type != JsonValue::Object) throw SceneError(C2RT_ERR_PARSE, "Invalid JSON in scene file!");
        return loadFromAbstractDataFormat(*makeVal(&root), baseDir);
    }
    if (ext == ".sdl") {
        const std::vector<SdlTag> tags = parseSdlSource(data);
        if (tags.empty()) throw SceneError(C2RT_ERR_PARSE, "Invalid SDL in scene file! (no root tag)");
        return loadFromAbstractDataFormat(*makeVal(&tags[0]), baseDir);
    }
    throw SceneError(C2RT_ERR_PARSE, "Error loading scene: unknown file type!");
}

std::unique_ptr<Scene> parseSceneFromFile(const std::string &filename)
{
    std::string data;
    if (!readFile(filename, data)) throw SceneError(C2RT_ERR_IO, "SceneNotFoundException: " + filename);
    return parseSceneFromString(data, extension(filename), dirName(filename));
}

// ------------------------------------------------------------------ flatten
const c2rt_scene_desc *Scene::flatten()
{
    flat_.reset(new Flat());
    Flat &f = *flat_;
    for (const Geometry &g : geometries) {
        f.geom_type.push_back(g.type);
        if (g.type == C2RT_GEOM_PLANE) { f.geom_param.insert(f.geom_param.end(), {g.y, g.limit, 0.0, 0.0}); }
        else if (g.type == C2RT_GEOM_SPHERE) { f.geom_param.insert(f.geom_param.end(), {g.center.x, g.center.y, g.center.z, g.R}); }
        else if (g.type == C2RT_GEOM_CUBE) { f.geom_param.insert(f.geom_param.end(), {g.center.x, g.center.y, g.center.z, g.side}); }
        else { f.geom_param.insert(f.geom_param.end(), {0.0, 0.0, 0.0, 0.0}); }
        f.geom_child.push_back(g.left);
        f.geom_child.push_back(g.right);
    }
    uint64_t offset = 0;
    for (const Texture &t : textures) {
        f.tex_type.push_back(t.type);
        float col[18] = {0};
        double par[6] = {0};
        if (t.type == C2RT_TEX_CHECKER) {
            const float c6[6] = {t.color1.r, t.color1.g, t.color1.b, t.color2.r, t.color2.g, t.color2.b};
            std::memcpy(col, c6, sizeof c6);
            par[0] = t.size;
        } else if (t.type == C2RT_TEX_PROCEDURE2) {
            for (int i = 0; i < 3; ++i) {
                col[3 * i + 0] = t.colorU[i].r; col[3 * i + 1] = t.colorU[i].g; col[3 * i + 2] = t.colorU[i].b;
                col[9 + 3 * i + 0] = t.colorV[i].r; col[9 + 3 * i + 1] = t.colorV[i].g; col[9 + 3 * i + 2] = t.colorV[i].b;
                par[i] = t.freqU[i];
                par[3 + i] = t.freqV[i];
            }
        }
        f.tex_color.insert(f.tex_color.end(), col, col + 18);
        f.tex_param.insert(f.tex_param.end(), par, par + 6);
        f.tex_scaling.push_back(t.scaling);
        f.tex_width.push_back(t.type == C2RT_TEX_BITMAP ? t.bmp.width : 0);
        f.tex_height.push_back(t.type == C2RT_TEX_BITMAP ? t.bmp.height : 0);
        f.tex_offset.push_back(offset);
        if (t.type == C2RT_TEX_BITMAP) {
            f.texels.insert(f.texels.end(), t.bmp.pixels.begin(), t.bmp.pixels.end());
            offset += (uint64_t)t.bmp.width * t.bmp.height;
        }
    }
    for (const Shader &s : shaders) {
        f.shader_type.push_back(s.type);
        f.shader_color.insert(f.shader_color.end(), {s.color.r, s.color.g, s.color.b});
        f.shader_texture.push_back(s.texture);
        f.shader_exponent.push_back(s.exponent);
        f.shader_strength.push_back(s.strength);
    }
    for (const Light &l : lights) {
        f.light_type.push_back(l.type);
        f.light_pos.insert(f.light_pos.end(), {l.pos.x, l.pos.y, l.pos.z});
        f.light_color.insert(f.light_color.end(), {l.lightColor.r, l.lightColor.g, l.lightColor.b});
        f.light_power.push_back(l.lightPower);
    }
    for (const Node &n : nodes) {
        f.node_geom.push_back(n.geom);
        f.node_shader.push_back(n.shader);
        f.node_bump.push_back(n.bumpmap);
        const Matrix *ms[3] = {&n.transform.transform, &n.transform.inverseTransform, &n.transform.transposedInverse};
        for (const Matrix *m : ms)
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) f.node_transform.push_back(m->c[i][j]);
        f.node_transform.insert(f.node_transform.end(), {n.transform.offset.x, n.transform.offset.y, n.transform.offset.z});
    }
    c2rt_scene_desc &d = f.desc;
    std::memset(&d, 0, sizeof d);
    d.abi_version = C2RT_ABI_VERSION;
    d.n_geoms = (uint32_t)geometries.size();
    d.geom_type = f.geom_type.data();
    d.geom_param = f.geom_param.data();
    d.geom_child = f.geom_child.data();
    d.n_textures = (uint32_t)textures.size();
    d.tex_type = f.tex_type.data();
    d.tex_color = f.tex_color.data();
    d.tex_param = f.tex_param.data();
    d.tex_scaling = f.tex_scaling.data();
    d.tex_width = f.tex_width.data();
    d.tex_height = f.tex_height.data();
    d.tex_offset = f.tex_offset.data();
    d.n_texels = offset;
    d.texels = f.texels.data();
    d.n_shaders = (uint32_t)shaders.size();
    d.shader_type = f.shader_type.data();
    d.shader_color = f.shader_color.data();
    d.shader_texture = f.shader_texture.data();
    d.shader_exponent = f.shader_exponent.data();
    d.shader_strength = f.shader_strength.data();
    d.n_lights = (uint32_t)lights.size();
    d.light_type = f.light_type.data();
    d.light_pos = f.light_pos.data();
    d.light_color = f.light_color.data();
    d.light_power = f.light_power.data();
    d.n_nodes = (uint32_t)nodes.size();
    d.node_geom = f.node_geom.data();
    d.node_shader = f.node_shader.data();
    d.node_bump = f.node_bump.data();
    d.node_transform = f.node_transform.data();
    d.ambient[0] = settings.ambientLightColor.r;
    d.ambient[1] = settings.ambientLightColor.g;
    d.ambient[2] = settings.ambientLightColor.b;
    d.max_trace_depth = settings.maxTraceDepth;
    d.gi_enabled = settings.GIEnabled ? 1u : 0u;
    return &d;
}

} // namespace host
} // namespace c2rt
