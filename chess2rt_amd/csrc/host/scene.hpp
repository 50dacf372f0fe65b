// scene.hpp — host-side mirror of the reference's scene graph (rt/*.d), kept
// to what the render hot path and its callers read.  Names follow the
// reference so that tests read like the reference's API.
#pragma once

#include <cmath>
#include <cstdint>
#include <limits>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../../include/c2rt.h"
#include "dsc.hpp"

namespace c2rt {
namespace host {

constexpr double kNaN = std::numeric_limits<double>::quiet_NaN();

struct Vector { // gfm vec3d (default-initialised to NaN like D doubles)
    double x = kNaN, y = kNaN, z = kNaN;
    Vector() = default;
    Vector(double x_, double y_, double z_) : x(x_), y(y_), z(z_) {}
};
struct Matrix { // gfm mat3d, row-major c[i][j]
    double c[3][3];
    static Matrix identity();
    static Matrix rotateX(double a);
    static Matrix rotateY(double a);
    static Matrix rotateZ(double a);
    Matrix operator*(const Matrix &b) const;
    Matrix inverse() const;
    Matrix transposed() const;
};
Vector mul(const Vector &v, const Matrix &m); // rt/imported_types.d:13-20
double radians(double deg);

struct Color { // rt/color.d:27-35
    float r = 0, g = 0, b = 0;
    Color() = default;
    Color(float r_, float g_, float b_) : r(r_), g(g_), b(b_) {}
};

// rt/global_settings.d:5-32
struct GlobalSettings {
    uint32_t frameWidth = 640, frameHeight = 480;
    bool fullscreen = false, allowResize = false, dynamicAspectRatio = false, interactive = false;
    uint32_t bucketSize = 48, threadCount = 0;
    bool prepassEnabled = true, prepassOnly = false, GIEnabled = false, AAEnabled = true;
    double AAThreshold = 0.1;
    uint32_t pathsPerPixel = 40, maxTraceDepth = 4;
    Color ambientLightColor;
    bool debugEnabled = true;
};

// rt/camera.d:11-256
struct Camera {
    uint64_t frameWidth = 0, frameHeight = 0;
    double aspect = 1.0;
    Vector pos;
    double yaw = 0, pitch = 0, roll = 0, fov = 0;
    double focalPlaneDist = 1.0, fNumber = 1.0, discMultiplier = kNaN;
    bool dof = false;
    uint64_t numSamples = 25;
    double stereoSeparation = 0;
    Vector upLeft, upRight, downLeft, frontDir, rightDir, upDir;

    void beginFrame();                                   // :77-117
    void move(double dx, double dy, double dz);          // :181-206
    void rotate(double dYaw, double dRoll, double dPitch); // :212-229
    void setFrameSize(uint32_t w, uint32_t h);           // :231-236
    void fill(c2rt_camera_frame &out) const;
};

// rt/transform.d
struct Transform {
    Matrix transform, inverseTransform, transposedInverse;
    Vector offset;
    void reset();
    void scale(double x, double y, double z);
    void rotate(double yaw, double pitch, double roll);
    void translate(const Vector &v);
};

struct Geometry { // rt/geometry.d
    int type = C2RT_GEOM_PLANE;
    double y = kNaN, limit = kNaN;          // Plane (both NaN after `this()`, :18-21)
    Vector center{0, 0, 0};                 // Sphere / Cube
    double R = 1, side = 1;
    int left = -1, right = -1;              // CsgOp children (indices into Scene::geometries)
    std::string name;
};

struct Bitmap { // rt/bitmap.d, imageio/image.d
    uint32_t width = 0, height = 0;
    std::vector<float> pixels; // width*height*3, y = 0 top
};

struct Texture { // rt/texture.d
    int type = C2RT_TEX_CHECKER;
    Color color1{0, 0, 0}, color2{1, 1, 1};
    double size = 1.0;
    std::vector<Color> colorU, colorV;
    std::vector<double> freqU, freqV;
    float scaling = 1, assumedGamma = 2.2f;
    Bitmap bmp;
    std::string name;
};

struct Shader { // rt/shader.d
    int type = C2RT_SHADER_LAMBERT;
    Color color{1, 1, 1};
    int texture = -1;
    double exponent = 16.0;
    float strength = 1.0f;
    std::string name;
};

struct Light { // rt/light.d
    int type = C2RT_LIGHT_POINT;
    Color lightColor;
    float lightPower = std::numeric_limits<float>::quiet_NaN();
    Vector pos;
    std::string name;
};

struct Node { // rt/node.d
    int geom = -1, shader = -1, bumpmap = -1;
    Transform transform;
    std::string name;
    Node() { transform.reset(); }
};

// rt/scene.d:39-96 + the flat view for c2rt_upload_scene
struct Scene {
    std::string name;
    GlobalSettings settings;
    Camera camera;
    std::vector<Light> lights;
    std::vector<Geometry> geometries;
    std::vector<Texture> textures;
    std::vector<Shader> shaders;
    std::vector<Node> nodes;

    void beginFrame() { camera.beginFrame(); } // rt/scene.d:55-58

    // Scene -> c2rt_scene_desc (tables owned by this object)
    const c2rt_scene_desc *flatten();

    // identity of the uploaded tables, so that Renderer uploads once per scene
    uint64_t upload_generation = 1;

private:
    struct Flat {
        std::vector<int32_t> geom_type, geom_child, tex_type, shader_type, shader_texture, light_type, node_geom,
            node_shader, node_bump;
        std::vector<double> geom_param, tex_param, shader_exponent, light_pos, node_transform;
        std::vector<float> tex_color, tex_scaling, texels, shader_color, shader_strength, light_color, light_power;
        std::vector<uint32_t> tex_width, tex_height;
        std::vector<uint64_t> tex_offset;
        c2rt_scene_desc desc;
    };
    std::unique_ptr<Flat> flat_;
};

// rt/scene_loader.d:20-41.  Throws SceneError (C2RT_ERR_IO / C2RT_ERR_PARSE).
std::unique_ptr<Scene> parseSceneFromFile(const std::string &filename);
// same, from memory (`ext` = ".sdl" or ".json"; `base_dir` resolves texture paths)
std::unique_ptr<Scene> parseSceneFromString(const std::string &data, const std::string &ext, const std::string &base_dir);

// imageio/bmp.d:60-193 + rt/color.d:60-66 (Color(uint)); throws SceneError
Bitmap loadBmpImage(const uint8_t *bytes, size_t len);
// rt/bitmap.d:116-136 dispatched as rt/texture.d:137-141
void applyAssumedGamma(float *texels, size_t n_floats, float assumedGamma);
// rt/color.d:154-162 via the cached table :209-228
uint32_t colorToRGB32(const float rgb[3]);
// imageio/bmp.d:195-237 (24-bpp, bottom-up, rows NOT padded: as written)
std::vector<uint8_t> saveBmp(const float *rgb, uint32_t width, uint32_t height);

} // namespace host
} // namespace c2rt
