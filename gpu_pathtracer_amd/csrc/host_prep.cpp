// host_prep.cpp — CPU-side preparation of what BeginRender uploads:
//   gpt_bvh_build           BVH::build / split / flatten   (reference src/bvh.cpp:18-173)
//   gpt_light_distribution  Scene::Init light power CDF    (reference src/scene.h:65-82)
//   gpt_infinite_init       Infinite::Init                 (reference src/infinite.h:61-63, src/bbox.h:98-101)
//   gpt_camera_init         Camera::Lookat + constructor   (reference src/camera.h:31-46,123-128)
//
// The BVH is the reference's structure, not a different one: binned SAH over
// primitive-bbox centroids, 12 buckets, leaf at <= 4 primitives or a bbox
// thinner than 1e-4, preorder flatten with leaf primitives appended in visit
// order.  Its topology and primitive order decide which hit wins a tie, so they
// are part of the results contract.  What changes
// is how it is built: per-primitive boxes and centroids are computed once and
// the recursion partitions an index array in place (the reference copies 176-B
// primitives into fresh vectors at every level).
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/gpt.h"
#include "host_util.h"
#include "pt_vec.h"

using namespace pt;

namespace {

struct Box {
    V3 lo{INFINITY, INFINITY, INFINITY};
    V3 hi{-INFINITY, -INFINITY, -INFINITY};
    void expand(V3 v)   // bbox.h:40-48
    {
        lo = V3{fmin_(lo.x, v.x), fmin_(lo.y, v.y), fmin_(lo.z, v.z)};
        hi = V3{fmax_(hi.x, v.x), fmax_(hi.y, v.y), fmax_(hi.z, v.z)};
    }
    void expand(const Box &b)   // bbox.h:30-38
    {
        lo = V3{fmin_(b.lo.x, lo.x), fmin_(b.lo.y, lo.y), fmin_(b.lo.z, lo.z)};
        hi = V3{fmax_(b.hi.x, hi.x), fmax_(b.hi.y, hi.y), fmax_(b.hi.z, hi.z)};
    }
    float surface_area() const   // bbox.h:62-65
    {
        V3 d = hi - lo;
        return 2.f * (d.x * d.y + d.y * d.z + d.z * d.x);
    }
};

float axis(V3 v, int a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }

struct Builder {
    const gpt_primitive *in;
    std::vector<Box> boxes;        // Triangle::GetBBox per input primitive (mesh.h:29-37)
    std::vector<V3> centers;       // BBox::Centric (bbox.h:50-52)
    std::vector<int> order;        // working permutation; subranges are node primitive lists
    std::vector<int> scratch;
    std::vector<gpt_bvh_node> *nodes;
    gpt_primitive *out;
    int n_out = 0;

    static constexpr int kBuckets = 12;

    int bucket_of(int prim, int ax, float start, float end) const
    {
        float value = axis(centers[(size_t)prim], ax);
        int no = (int)((value - start) / (end - start) * kBuckets);
        if (no == kBuckets) no = no - 1;                       // bvh.cpp:78
        return no < 0 ? 0 : (no > kBuckets - 1 ? kBuckets - 1 : no);   // (never taken for finite input: centroids lie inside the node box)
    }

    void emit_leaf(int node, int first, int count)
    {
        gpt_bvh_node &n = (*nodes)[(size_t)node];
        n.is_leaf = 1;
        n.second_child_offset = -1;
        if (count > 0) {
            n.start = n_out;
            for (int i = 0; i < count; ++i) out[n_out++] = in[order[(size_t)(first + i)]];
            n.end = n_out - 1;
        }
    }

    // Builds the subtree over order[first, first+count) directly in preorder:
    // node index = nodes->size() at entry (split() numbers nodes in the same
    // order flatten() later visits them: node, left subtree, right subtree).
    void build(int first, int count, const Box &bbox)
    {
        const int node = (int)nodes->size();
        gpt_bvh_node n;
        std::memset(&n, 0, sizeof(n));
        n.fmin = gpt_float3{bbox.lo.x, bbox.lo.y, bbox.lo.z};
        n.fmax = gpt_float3{bbox.hi.x, bbox.hi.y, bbox.hi.z};
        n.start = n.end = -1;
        nodes->push_back(n);

        V3 diagonal = bbox.hi - bbox.lo;
        if (count <= 4 || diagonal.x < 0.0001f || diagonal.y < 0.0001f || diagonal.z < 0.0001f) {
            emit_leaf(node, first, count);
            return;
        }
        // Binned SAH (bvh.cpp:55-106): 11 candidate planes per axis, cost = SA(left) * n_left + SA(right) * n_right, an
        // empty side costs 0, and a split must beat n * SA(parent); ties keep the earlier (axis, plane).  The reference
        // re-unites the buckets on both sides for every plane; here the right-hand unions come from one sweep from the top
        // and the left-hand ones grow with the plane.  min/max unions are exact and order-independent, so boxes, counts
        // and therefore every cost are the same floats.
        int split_axis = -1, split_plane = 0;
        float lowest = (float)(size_t)count * bbox.surface_area();
        for (int ax = 0; ax < 3; ++ax) {
            Box bin_box[kBuckets];
            int bin_n[kBuckets] = {0};
            const float lo = axis(bbox.lo, ax), hi = axis(bbox.hi, ax);
            for (int j = 0; j < count; ++j) {
                const int p = order[(size_t)(first + j)];
                const int b = bucket_of(p, ax, lo, hi);
                bin_n[b]++;
                bin_box[b].expand(boxes[(size_t)p]);
            }
            Box above[kBuckets];             // above[j] = union of bins j .. 11
            int n_above[kBuckets];
            above[kBuckets - 1] = bin_box[kBuckets - 1];
            n_above[kBuckets - 1] = bin_n[kBuckets - 1];
            for (int j = kBuckets - 2; j >= 1; --j) {
                above[j] = above[j + 1];
                above[j].expand(bin_box[j]);
                n_above[j] = n_above[j + 1] + bin_n[j];
            }
            Box below;                       // union of bins 0 .. j-1
            int n_below = 0;
            for (int j = 1; j < kBuckets; ++j) {
                below.expand(bin_box[j - 1]);
                n_below += bin_n[j - 1];
                const float left = (n_below == 0) ? 0 : below.surface_area() * n_below;
                const float right = (n_above[j] == 0) ? 0 : above[j].surface_area() * n_above[j];
                const float sah = left + right;
                if (sah < lowest) {
                    lowest = sah;
                    split_axis = ax;
                    split_plane = j;
                }
            }
        }
        const int best_axis = split_axis, best_bucket = split_plane;
        if (best_axis == -1) {
            emit_leaf(node, first, count);
            return;
        }
        // stable partition (the reference pushes into left/right vectors in input order)
        const float start = axis(bbox.lo, best_axis), end = axis(bbox.hi, best_axis);
        Box best_left, best_right;
        int nl = 0, nr = 0;
        for (int i = 0; i < count; ++i) {
            const int p = order[(size_t)(first + i)];
            if (bucket_of(p, best_axis, start, end) < best_bucket) {
                order[(size_t)(first + nl++)] = p;     // nl <= i: never overtakes the read cursor
                best_left.expand(boxes[(size_t)p]);
            } else {
                scratch[(size_t)nr++] = p;
                best_right.expand(boxes[(size_t)p]);
            }
        }
        std::memcpy(&order[(size_t)(first + nl)], scratch.data(), sizeof(int) * (size_t)nr);

        (*nodes)[(size_t)node].is_leaf = 0;
        build(first, nl, best_left);
        (*nodes)[(size_t)node].second_child_offset = (int)nodes->size();
        build(first + nl, nr, best_right);
    }
};

}  // namespace

extern "C" {

int gpt_bvh_build(const gpt_primitive *prims_in, int32_t n, gpt_primitive *prims_out, gpt_bvh_node *nodes_out,
                  int32_t *n_nodes_out, float root_box6[6])
{
    if (n < 0 || (n > 0 && (!prims_in || !prims_out || !nodes_out)) || !n_nodes_out) {
        gpt_set_error("gpt_bvh_build: invalid argument");
        return GPT_ERR_INVALID_ARG;
    }
    *n_nodes_out = 0;
    if (n == 0) return GPT_OK;
    for (int i = 0; i < n; ++i) {
        if (prims_in[i].type != GPT_GT_TRIANGLE) {
            gpt_set_error("gpt_bvh_build: primitive %d has type %d; only triangles are supported", i, prims_in[i].type);
            return GPT_ERR_UNSUPPORTED;
        }
    }
    Builder b;
    b.in = prims_in;
    b.out = prims_out;
    b.boxes.resize((size_t)n);
    b.centers.resize((size_t)n);
    b.order.resize((size_t)n);
    b.scratch.resize((size_t)n);
    Box root;
    for (int i = 0; i < n; ++i) {
        const gpt_triangle &t = prims_in[i].triangle;
        const float c9[9] = {t.v1.v.x, t.v1.v.y, t.v1.v.z, t.v2.v.x, t.v2.v.y, t.v2.v.z, t.v3.v.x, t.v3.v.y, t.v3.v.z};
        for (float c : c9) {
            if (!std::isfinite(c)) {      // the reference's bucket index is undefined for such input (bvh.cpp:77)
                gpt_set_error("gpt_bvh_build: primitive %d has a non-finite vertex coordinate", i);
                return GPT_ERR_INVALID_ARG;
            }
        }
        Box bx;
        bx.expand(V3{t.v1.v.x, t.v1.v.y, t.v1.v.z});
        bx.expand(V3{t.v2.v.x, t.v2.v.y, t.v2.v.z});
        bx.expand(V3{t.v3.v.x, t.v3.v.y, t.v3.v.z});
        b.boxes[(size_t)i] = bx;
        b.centers[(size_t)i] = (bx.lo + bx.hi) * 0.5f;
        b.order[(size_t)i] = i;
        root.expand(bx);
    }
    std::vector<gpt_bvh_node> nodes;
    nodes.reserve((size_t)n);
    b.nodes = &nodes;
    b.build(0, n, root);
    std::memcpy(nodes_out, nodes.data(), nodes.size() * sizeof(gpt_bvh_node));
    *n_nodes_out = (int32_t)nodes.size();
    if (root_box6) {
        root_box6[0] = root.lo.x; root_box6[1] = root.lo.y; root_box6[2] = root.lo.z;
        root_box6[3] = root.hi.x; root_box6[4] = root.hi.y; root_box6[5] = root.hi.z;
    }
    return GPT_OK;
}

int gpt_light_distribution(const gpt_area *lights, int32_t n_lights, const gpt_infinite *infinite, float *cdf_out,
                           int32_t *n_out)
{
    if (n_lights < 0 || (n_lights > 0 && !lights) || !cdf_out || !n_out) {
        gpt_set_error("gpt_light_distribution: invalid argument");
        return GPT_ERR_INVALID_ARG;
    }
    const V3 luma = v3(0.212671f, 0.715160f, 0.072169f);
    float sum = 0.f;
    int n = 0;
    cdf_out[n++] = 0.f;
    for (int i = 0; i < n_lights; ++i) {
        const gpt_triangle &t = lights[i].triangle;
        const V3 v1 = V3{t.v1.v.x, t.v1.v.y, t.v1.v.z};
        const V3 e1 = V3{t.v2.v.x, t.v2.v.y, t.v2.v.z} - v1;
        const V3 e2 = V3{t.v3.v.x, t.v3.v.y, t.v3.v.z} - v1;
        const float area = length(cross(e1, e2)) * 0.5f;                        // mesh.h:39-43
        const V3 radiance = V3{lights[i].radiance.x, lights[i].radiance.y, lights[i].radiance.z};
        const V3 power = radiance * area * PI;                                   // area.h:34-36
        sum += dot(luma, power);
        cdf_out[n++] = sum;
    }
    if (infinite && infinite->isvalid) {
        if (!infinite->data) {
            gpt_set_error("gpt_light_distribution: infinite light has no data");
            return GPT_ERR_INVALID_ARG;
        }
        const V3 texel0 = V3{infinite->data[0].x, infinite->data[0].y, infinite->data[0].z};
        const V3 power = FOURPI * infinite->radius * infinite->radius * texel0;   // infinite.h:43-45
        sum += dot(luma, power);
        cdf_out[n++] = sum;
    }
    for (int i = 0; i < n; ++i) cdf_out[i] /= sum;
    *n_out = n;
    return GPT_OK;
}

int gpt_infinite_init(gpt_infinite *infinite, const float root_box6[6])
{
    if (!infinite || !root_box6) {
        gpt_set_error("gpt_infinite_init: null argument");
        return GPT_ERR_INVALID_ARG;
    }
    const V3 lo = V3{root_box6[0], root_box6[1], root_box6[2]};
    const V3 hi = V3{root_box6[3], root_box6[4], root_box6[5]};
    const V3 center = (lo + hi) * 0.5f;
    infinite->center = gpt_float3{center.x, center.y, center.z};
    infinite->radius = sqrt_rn(dot(hi - center, hi - center));
    return GPT_OK;
}

int gpt_camera_init(gpt_camera *c, const float position[3], const float lookat[3], const float up[3], float res_x,
                    float res_y, float distance, float fov_degrees, float aperture_radius, float focal_distance,
                    int filmic, int environment)
{
    if (!c || !position || !lookat || !up || res_x <= 0 || res_y <= 0) {
        gpt_set_error("gpt_camera_init: invalid argument");
        return GPT_ERR_INVALID_ARG;
    }
    std::memset(c, 0, sizeof(*c));
    const V3 eye = V3{position[0], position[1], position[2]};
    const V3 dest = V3{lookat[0], lookat[1], lookat[2]};
    const V3 upv = V3{up[0], up[1], up[2]};
    const V3 w = normalize(eye - dest);               // camera.h:123-128
    const V3 u = normalize(cross(upv, w));
    const V3 v = normalize(cross(w, u));
    c->position = gpt_float3{eye.x, eye.y, eye.z};
    c->u = gpt_float3{u.x, u.y, u.z};
    c->v = gpt_float3{v.x, v.y, v.z};
    c->w = gpt_float3{w.x, w.y, w.z};
    c->resolution.x = res_x;                          // camera.h:31-46
    c->resolution.y = res_y;
    c->distance = distance;
    c->fov = fov_degrees;
    c->apertureRadius = aperture_radius;
    c->focalDistance = focal_distance;
    c->filmic = filmic ? 1 : 0;
    c->environment = environment ? 1 : 0;
    c->medium = -1;
    const float half_fov = fov_degrees * .5f;
    const float radians = (float)(half_fov / 180.0 * PI);   // DegreesToRadians, common.h:46-49 (double arithmetic)
    c->height = std::tan(radians) * distance;                // float overload == tanf
    c->width = c->height * res_x / res_y;
    c->area = 4.f * c->width * c->height;
    c->pixel2screen.x = 2.f * c->width / res_x;
    c->pixel2screen.y = 2.f * c->height / res_y;
    c->ratio = focal_distance / distance;
    return GPT_OK;
}

}  // extern "C"
