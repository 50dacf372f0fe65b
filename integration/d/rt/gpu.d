/**
 * rt.gpu — D-side binding of libc2rt.so (include/c2rt.h) for Chess2RT.
 *
 * WRITTEN, NOT COMPILED: the build environment of this repository has no D
 * toolchain (no dmd / ldc2 / gdc / dub), so this file could not be compiled
 * or run.  The identical boundary is exercised from C++
 * (chess2rt_amd/csrc/host/host_api.cpp + scene.cpp: `Scene::flatten`,
 * `Renderer::renderRT`) and from Python (chess2rt_amd/_abi.py), which is
 * what the tests use.  Drop this file into `source/rt/`, add `libs "c2rt"`
 * to dub.sdl and replace the lambda body in rt/renderer.d:36-37 as shown at
 * the bottom.
 *
 * Access note: Camera keeps upLeft/upRight/downLeft/frontDir/rightDir/upDir,
 * Sphere/CsgOp/BitmapTexture/Transform keep their fields `private` or
 * `protected` (rt/camera.d:47-53, rt/geometry.d:75-79,253-257,
 * rt/texture.d:149-161, rt/transform.d:11-14).  D's `private` is
 * module-level, so either move `flatten` pieces into those modules or relax
 * the attributes to `package`; the code below assumes `package`.
 */
module rt.gpu;

import core.atomic : atomicStore;
import std.exception : enforce;
import std.string : fromStringz;

import imageio.image : Image;
import rt.camera, rt.color, rt.geometry, rt.light, rt.node, rt.scene, rt.shader, rt.texture, rt.importedtypes;

extern (C) nothrow @nogc
{
    struct c2rt_ctx;

    enum : int { C2RT_OK = 0, C2RT_ERR_CANCELLED = 7 }
    enum : int { GEOM_PLANE, GEOM_SPHERE, GEOM_CUBE, GEOM_CSG_UNION, GEOM_CSG_INTER, GEOM_CSG_DIFF }
    enum : int { SHADER_LAMBERT, SHADER_PHONG }
    enum : int { TEX_CHECKER, TEX_PROCEDURE2, TEX_BITMAP }

    struct c2rt_scene_desc
    {
        uint abi_version = 1;
        uint n_geoms; const(int)* geom_type; const(double)* geom_param; const(int)* geom_child;
        uint n_textures; const(int)* tex_type; const(float)* tex_color; const(double)* tex_param;
        const(float)* tex_scaling; const(uint)* tex_width; const(uint)* tex_height; const(ulong)* tex_offset;
        ulong n_texels; const(float)* texels;
        uint n_shaders; const(int)* shader_type; const(float)* shader_color; const(int)* shader_texture;
        const(double)* shader_exponent; const(float)* shader_strength;
        uint n_lights; const(int)* light_type; const(double)* light_pos; const(float)* light_color; const(float)* light_power;
        uint n_nodes; const(int)* node_geom; const(int)* node_shader; const(int)* node_bump; const(double)* node_transform;
        float[3] ambient; uint max_trace_depth; uint gi_enabled;
    }

    struct c2rt_camera_frame
    {
        double[3] pos, up_left, up_right, down_left, right_dir, up_dir, front_dir;
        double frame_width, frame_height;
        uint dof, num_samples;
        double focal_plane_dist, disc_multiplier, stereo_separation;
    }

    struct c2rt_render_opts
    {
        uint width, height, taps, strip_height, strip_rank, strip_world;
        ulong seed;
        uint count_rays, prepass_bucket;
    }

    struct c2rt_trace_result
    {
        float[3] color; int closest_node, leaf_geom;
        double[3] p, normal; double dist, u, v; double[3] ray_orig, ray_dir;
    }

    int c2rt_init(int device, c2rt_ctx** outCtx);
    int c2rt_init_multi(int device_count_or_0, const(int)* device_ids, c2rt_ctx** outCtx);
    int c2rt_device_count(const c2rt_ctx*);
    ulong c2rt_scene_generation(const c2rt_ctx*);
    void c2rt_destroy(c2rt_ctx*);
    const(char)* c2rt_last_error(const c2rt_ctx*);
    int c2rt_upload_scene(c2rt_ctx*, const c2rt_scene_desc*);
    int c2rt_render_frame(c2rt_ctx*, const c2rt_camera_frame*, const c2rt_render_opts*, float* out_rgb,
                          const shared(ubyte)* stop_flag);
    int c2rt_pin_host_buffer(c2rt_ctx*, float* out_rgb, size_t bytes);
    int c2rt_unpin_host_buffer(c2rt_ctx*, float* out_rgb);
    int c2rt_render_pixel(c2rt_ctx*, const c2rt_camera_frame*, const c2rt_render_opts*, int x, int y,
                          c2rt_trace_result*);
    /// display words (Color.toRGB32, encoded in the render kernel) straight into `screen`'s uint twin: what
    /// SDL2Gui.draw (gui/sdl2_gui.d:139-155) computes per pixel on the CPU today
    int c2rt_render_frame_rgb32(c2rt_ctx*, const c2rt_camera_frame*, const c2rt_render_opts*, uint* out_rgb32,
                                const shared(ubyte)* stop_flag);
    /// tiles rendered a second time through the IEEE divide / sqrt path (a cost indicator; 0 on sane scenes)
    int c2rt_get_exact_redos(c2rt_ctx*, ulong* outCount);
}

/// Owns the flat tables for one uploaded scene (GC memory; c2rt_upload_scene copies them).
final class GpuScene
{
    c2rt_scene_desc desc;
    private
    {
        int[] geomType, geomChild, texType, shaderType, shaderTexture, lightType, nodeGeom, nodeShader, nodeBump;
        double[] geomParam, texParam, shaderExponent, lightPos, nodeTransform;
        float[] texColor, texScaling, texels, shaderColor, shaderStrength, lightColor, lightPower;
        uint[] texWidth, texHeight;
        ulong[] texOffset;
    }

    private static int indexOf(T)(const T[] all, const Object o)
    {
        foreach (i, e; all) if (e is o) return cast(int) i;
        return -1;
    }

    /// Scene -> c2rt_scene_desc: class references become indices into Scene's arrays.
    this(const Scene scene)
    {
        foreach (g; scene.geometries)
        {
            if (auto p = cast(const Plane) g) { geomType ~= GEOM_PLANE; geomParam ~= [p.y, p.limit, 0, 0]; geomChild ~= [-1, -1]; }
            else if (auto s = cast(const Sphere) g) { geomType ~= GEOM_SPHERE; geomParam ~= [s.center.x, s.center.y, s.center.z, s.R]; geomChild ~= [-1, -1]; }
            else if (auto c = cast(const Cube) g) { geomType ~= GEOM_CUBE; geomParam ~= [c.center.x, c.center.y, c.center.z, c.side]; geomChild ~= [-1, -1]; }
            else if (auto op = cast(const CsgOp) g)
            {
                geomType ~= (cast(const CsgUnion) g) ? GEOM_CSG_UNION : (cast(const CsgInter) g) ? GEOM_CSG_INTER : GEOM_CSG_DIFF;
                geomParam ~= [0.0, 0, 0, 0];
                geomChild ~= [indexOf(scene.geometries, op.left.get), indexOf(scene.geometries, op.right.get)];
            }
        }
        ulong offset = 0;
        foreach (t; scene.textures)
        {
            float[18] col = 0; double[6] par = 0;
            uint w = 0, h = 0; float scaling = 1;
            if (auto c = cast(const Checker) t)
            {
                texType ~= TEX_CHECKER;
                col[0 .. 3] = c.color1.components; col[3 .. 6] = c.color2.components; par[0] = c.size;
            }
            else if (auto p = cast(const Procedure2) t)
            {
                texType ~= TEX_PROCEDURE2;
                foreach (i; 0 .. 3) { col[3*i .. 3*i+3] = p.colorU[i].components; col[9+3*i .. 12+3*i] = p.colorV[i].components;
                                      par[i] = p.freqU[i]; par[3+i] = p.freqV[i]; }
            }
            else if (auto b = cast(const BitmapTexture) t)
            {
                texType ~= TEX_BITMAP; scaling = b.scaling;
                w = cast(uint) b.bmp.width; h = cast(uint) b.bmp.height;
                foreach (px; b.bmp.data.pixels) texels ~= px.components;   // already linear RGB (decompressGamma_sRGB)
            }
            texColor ~= col; texParam ~= par; texScaling ~= scaling; texWidth ~= w; texHeight ~= h; texOffset ~= offset;
            offset += cast(ulong) w * h;
        }
        foreach (s; scene.shaders)
        {
            shaderColor ~= s.color.components;
            if (auto p = cast(const Phong) s)
            { shaderType ~= SHADER_PHONG; shaderTexture ~= indexOf(scene.textures, p.texture.get); shaderExponent ~= p.exponent; shaderStrength ~= p.strength; }
            else
            { auto l = cast(const Lambert) s; shaderType ~= SHADER_LAMBERT; shaderTexture ~= indexOf(scene.textures, l.texture.get); shaderExponent ~= 16; shaderStrength ~= 1; }
        }
        foreach (l; scene.lights)
        {
            auto p = cast(const PointLight) l;
            lightType ~= 0; lightPos ~= [p.pos.x, p.pos.y, p.pos.z]; lightColor ~= l.lightColor.components; lightPower ~= l.lightPower;
        }
        foreach (n; scene.nodes)
        {
            nodeGeom ~= indexOf(scene.geometries, n.geom); nodeShader ~= indexOf(scene.shaders, n.shader);
            nodeBump ~= indexOf(scene.textures, n.bumpmap);
            // gfm mat3d stores c[i][j] row-major in v[0 .. 9]
            nodeTransform ~= n.transform.transform.v[] ~ n.transform.inverseTransform.v[] ~ n.transform.transposedInverse.v[]
                           ~ [n.transform.offset.x, n.transform.offset.y, n.transform.offset.z];
        }
        with (desc)
        {
            n_geoms = cast(uint) geomType.length; geom_type = geomType.ptr; geom_param = geomParam.ptr; geom_child = geomChild.ptr;
            n_textures = cast(uint) texType.length; tex_type = texType.ptr; tex_color = texColor.ptr; tex_param = texParam.ptr;
            tex_scaling = texScaling.ptr; tex_width = texWidth.ptr; tex_height = texHeight.ptr; tex_offset = texOffset.ptr;
            n_texels = offset; texels = this.texels.ptr;
            n_shaders = cast(uint) shaderType.length; shader_type = shaderType.ptr; shader_color = shaderColor.ptr;
            shader_texture = shaderTexture.ptr; shader_exponent = shaderExponent.ptr; shader_strength = shaderStrength.ptr;
            n_lights = cast(uint) lightType.length; light_type = lightType.ptr; light_pos = lightPos.ptr;
            light_color = lightColor.ptr; light_power = lightPower.ptr;
            n_nodes = cast(uint) nodeGeom.length; node_geom = nodeGeom.ptr; node_shader = nodeShader.ptr; node_bump = nodeBump.ptr;
            node_transform = nodeTransform.ptr;
            ambient = scene.settings.ambientLightColor.components;
            max_trace_depth = scene.settings.maxTraceDepth;
            gi_enabled = scene.settings.GIEnabled;
        }
    }
}

/// The six vectors Camera.beginFrame leaves behind (rt/camera.d:77-117) + what getScreenRay reads.
c2rt_camera_frame frameOf(const Camera c)
{
    c2rt_camera_frame f;
    f.pos = c.pos.v; f.up_left = c.upLeft.v; f.up_right = c.upRight.v; f.down_left = c.downLeft.v;
    f.right_dir = c.rightDir.v; f.up_dir = c.upDir.v; f.front_dir = c.frontDir.v;
    f.frame_width = c.frameWidth; f.frame_height = c.frameHeight;
    f.dof = c.dof; f.num_samples = cast(uint) c.numSamples;
    f.focal_plane_dist = c.focalPlaneDist; f.disc_multiplier = c.discMultiplier; f.stereo_separation = c.stereoSeparation;
    return f;
}

/// One per process; created in RTDemo.init, scene uploaded in RTDemo.resetScene after parseSceneFromFile.
final class GpuRenderer
{
    private c2rt_ctx* ctx;
    private GpuScene uploaded;
    private float* pinned;   // the screen buffer currently page-locked through this context

    /// device >= 0: that GPU; device < 0: the current one; allDevices: ONE context over every visible
    /// GPU (c2rt_init_multi(0, null)): frames are dealt to the GPUs in interleaved 8-row strips and every
    /// GPU copies its strips straight into `output.pixels` over its own PCIe link.
    this(int device = -1, bool allDevices = false)
    {
        auto st = allDevices ? c2rt_init_multi(0, null, &ctx) : c2rt_init(device, &ctx);
        enforce(st == C2RT_OK, "no usable GPU");
    }
    ~this()
    {
        if (ctx && pinned) c2rt_unpin_host_buffer(ctx, pinned);
        if (ctx) c2rt_destroy(ctx);
    }

    void upload(const Scene scene)
    {
        uploaded = new GpuScene(scene);
        enforce(c2rt_upload_scene(ctx, &uploaded.desc) == C2RT_OK, c2rt_last_error(ctx).fromStringz);
    }

    /// `screen.alloc` happened (gui/raytracer_demo.d:181-182): let frames stream back at PCIe rate.
    /// The GUI re-allocates `screen` on every resize (gui/raytracer_demo.d:135-143,181-182): the previous
    /// buffer is unpinned first (its pages may already be back with the GC), then the new one is registered.
    void pin(Image!Color screen)
    {
        auto p = cast(float*) screen.pixels.ptr;
        if (pinned && pinned !is p) { c2rt_unpin_host_buffer(ctx, pinned); pinned = null; }
        if (c2rt_pin_host_buffer(ctx, p, screen.pixels.length * Color.sizeof) == C2RT_OK) pinned = p;
    }
    /// Call before the GUI frees or re-allocates the buffer passed to pin().
    void unpin() { if (pinned) { c2rt_unpin_host_buffer(ctx, pinned); pinned = null; } }

    /// Drop-in for `Renderer(scene, output, isRendering, isStopRequested).renderRT()`; call after scene.beginFrame().
    int renderRT(const Scene scene, Image!Color output, shared(bool)* isRendering, const shared(bool)* isStopRequested)
    {
        auto cam = frameOf(scene.camera);
        auto opts = c2rt_render_opts(scene.settings.frameWidth, scene.settings.frameHeight, scene.settings.AAEnabled ? 5 : 1);
        int st = c2rt_render_frame(ctx, &cam, &opts, cast(float*) output.pixels.ptr, cast(const shared(ubyte)*) isStopRequested);
        if (isRendering !is null) (*isRendering).atomicStore(false);   // end(), rt/renderer.d:87-91
        return st;
    }
}

/+ rt/renderer.d, renderSceneAsync — only the lambda body changes:

    scene.beginFrame();
    spawn((shared Scene s, shared Image!Color o, shared(bool)* isWorking, const shared(bool)* isStopping)
        {
            gpuRenderer.renderRT(cast() s, cast() o, isWorking, isStopping);   // was: Renderer(...).renderRT()
        }, cast(shared) scene, cast(shared) output, isRendering, needsRendering);
+/
