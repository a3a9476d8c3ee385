// ref_transform.cpp — TEST INFRASTRUCTURE: the reference's transform arithmetic, through the glm it vendors.
//
// The reference places meshes and turns its environment light with glm (header-only, /root/reference/include/glm,
// version 0.9): src/parsescene.cpp:349-355 builds t * r * s from "scale" / "translate" / "rotate", src/mesh.cpp:49-57
// moves every vertex by it and every normal by the inverse transpose, src/parsescene.cpp:552-569 derives the
// environment light's u, v, w from "rotate" or "matrix".  glm compiles with g++ as it lies, so those exact calls are
// made here, behind a C ABI for tests/ (oracle/_ref/libref_transform.so, oracle/Makefile).  No glm code is copied;
// nothing in the product links or calls this.
#include <cstring>
#include <glm/glm.hpp>
#include <glm/gtc/matrix_transform.hpp>

using namespace glm;
#define REF_API extern "C" __attribute__((visibility("default")))

// src/parsescene.cpp:349-355 (and :402-408, :508-514): column-major 4x4 out
REF_API void ref_mesh_trs(const float scale3[3], const float translate3[3], const float rotate3[3], float out16[16])
{
    mat4 trs, t, r, s;
    s = glm::scale(s, vec3(scale3[0], scale3[1], scale3[2]));
    t = glm::translate(t, vec3(translate3[0], translate3[1], translate3[2]));
    r = glm::rotate(r, radians(rotate3[0]), vec3(1, 0, 0));
    r = glm::rotate(r, radians(rotate3[1]), vec3(0, 1, 0));
    r = glm::rotate(r, radians(rotate3[2]), vec3(0, 0, 1));
    trs = t * r * s;
    memcpy(out16, &trs[0], 16 * sizeof(float));
}

// src/mesh.cpp:49-57
REF_API void ref_transform_vertices(const float trs16[16], int n, const float *v_in, const float *n_in, float *v_out, float *n_out)
{
    mat4 trs;
    memcpy(&trs[0], trs16, 16 * sizeof(float));
    mat4 invT = transpose(inverse(trs));
    for (int i = 0; i < n; ++i) {
        vec3 v(v_in[3 * i], v_in[3 * i + 1], v_in[3 * i + 2]);
        vec3 nn(n_in[3 * i], n_in[3 * i + 1], n_in[3 * i + 2]);
        v = vec3(trs * vec4(v, 1));
        nn = normalize(vec3(invT * vec4(nn, 0)));
        v_out[3 * i] = v.x; v_out[3 * i + 1] = v.y; v_out[3 * i + 2] = v.z;
        n_out[3 * i] = nn.x; n_out[3 * i + 1] = nn.y; n_out[3 * i + 2] = nn.z;
    }
}

// src/parsescene.cpp:552-561 ("rotate") and :563-569 ("matrix": 16 numbers copied into the matrix, then inverted)
REF_API void ref_infinite_frame(const float *rotate3, const float *matrix16, float uvw9[9])
{
    vec3 uu, vv, ww;
    if (rotate3) {
        mat4 rs;
        rs = rotate(rs, radians(rotate3[0]), vec3(1, 0, 0));
        rs = rotate(rs, radians(rotate3[1]), vec3(0, 1, 0));
        rs = rotate(rs, radians(rotate3[2]), vec3(0, 0, 1));
        uu = vec3(rs * vec4(1, 0, 0, 0));
        vv = vec3(rs * vec4(0, 1, 0, 0));
        ww = vec3(rs * vec4(0, 0, 1, 0));
    }
    if (matrix16) {
        mat4 rs;
        memcpy(&rs[0], matrix16, 16 * sizeof(float));
        rs = inverse(rs);
        uu = vec3(rs * vec4(1, 0, 0, 0));
        vv = vec3(rs * vec4(0, 1, 0, 0));
        ww = vec3(rs * vec4(0, 0, 1, 0));
    }
    uvw9[0] = uu.x; uvw9[1] = uu.y; uvw9[2] = uu.z;
    uvw9[3] = vv.x; uvw9[4] = vv.y; uvw9[5] = vv.z;
    uvw9[6] = ww.x; uvw9[7] = ww.y; uvw9[8] = ww.z;
}
