// pt_hostmath.h -- host-side matrix/camera arithmetic of the scene loader.
//
// Restates what the reference gets from glm 0.9.6.3 (apps/external/include/glm) through
// utilityCore::buildTransformationMatrix (src/utilities.cpp:65-72), glm::inverse / glm::inverseTranspose
// (src/scene.cpp:301-304), Scene::loadCamera (src/scene.cpp:364-374) and runCuda (src/main.cpp:56-70, 105-123),
// operation by operation so that the 3 matrices per geom and the camera come out bit-identical.  Host only:
// transcendental functions are glibc's float versions, exactly what the reference's loader calls.
#pragma once
#include <cmath>
#include <cstring>

namespace pth {

struct Mat4 { float m[16]; };      // glm memory order: m[c*4 + r]

inline Mat4 identity() { Mat4 r; std::memset(r.m, 0, sizeof r.m); r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.f; return r; }

// glm operator*(mat4, mat4): Result[c] = A[0]*B[c][0] + A[1]*B[c][1] + A[2]*B[c][2] + A[3]*B[c][3]
inline Mat4 mul(const Mat4 &A, const Mat4 &B) {
    Mat4 R;
    for (int c = 0; c < 4; c++)
        for (int r = 0; r < 4; r++) {
            float acc = A.m[0 * 4 + r] * B.m[c * 4 + 0] + A.m[1 * 4 + r] * B.m[c * 4 + 1];
            acc = acc + A.m[2 * 4 + r] * B.m[c * 4 + 2];
            acc = acc + A.m[3 * 4 + r] * B.m[c * 4 + 3];
            R.m[c * 4 + r] = acc;
        }
    return R;
}

// glm::translate (gtc/matrix_transform.inl:40-50)
inline Mat4 translate(const Mat4 &M, const float v[3]) {
    Mat4 R = M;
    for (int r = 0; r < 4; r++) {
        float acc = M.m[0 + r] * v[0] + M.m[4 + r] * v[1];
        acc = acc + M.m[8 + r] * v[2];
        R.m[12 + r] = acc + M.m[12 + r];
    }
    return R;
}

// glm::rotate (gtc/matrix_transform.inl:52-84)
inline Mat4 rotate(const Mat4 &M, float angle, const float axis_in[3]) {
    const float c = cosf(angle), s = sinf(angle);
    const float inv = 1.0f / sqrtf(axis_in[0] * axis_in[0] + axis_in[1] * axis_in[1] + axis_in[2] * axis_in[2]);
    const float ax[3] = {axis_in[0] * inv, axis_in[1] * inv, axis_in[2] * inv};
    const float tmp[3] = {(1.f - c) * ax[0], (1.f - c) * ax[1], (1.f - c) * ax[2]};
    float Rt[3][3];
    Rt[0][0] = c + tmp[0] * ax[0];
    Rt[0][1] = 0 + tmp[0] * ax[1] + s * ax[2];
    Rt[0][2] = 0 + tmp[0] * ax[2] - s * ax[1];
    Rt[1][0] = 0 + tmp[1] * ax[0] - s * ax[2];
    Rt[1][1] = c + tmp[1] * ax[1];
    Rt[1][2] = 0 + tmp[1] * ax[2] + s * ax[0];
    Rt[2][0] = 0 + tmp[2] * ax[0] + s * ax[1];
    Rt[2][1] = 0 + tmp[2] * ax[1] - s * ax[0];
    Rt[2][2] = c + tmp[2] * ax[2];
    Mat4 R;
    for (int j = 0; j < 3; j++)
        for (int r = 0; r < 4; r++) {
            float acc = M.m[0 + r] * Rt[j][0] + M.m[4 + r] * Rt[j][1];
            R.m[j * 4 + r] = acc + M.m[8 + r] * Rt[j][2];
        }
    for (int r = 0; r < 4; r++) R.m[12 + r] = M.m[12 + r];
    return R;
}

// glm::scale (gtc/matrix_transform.inl:120-133)
inline Mat4 scale(const Mat4 &M, const float v[3]) {
    Mat4 R;
    for (int r = 0; r < 4; r++) {
        R.m[0 + r] = M.m[0 + r] * v[0];
        R.m[4 + r] = M.m[4 + r] * v[1];
        R.m[8 + r] = M.m[8 + r] * v[2];
        R.m[12 + r] = M.m[12 + r];
    }
    return R;
}

// glm::inverse(mat4) = detail::compute_inverse (detail/type_mat4x4.inl:36-89)
inline Mat4 inverse(const Mat4 &A) {
    auto e = [&](int c, int r) { return A.m[c * 4 + r]; };
    const float c00 = e(2,2) * e(3,3) - e(3,2) * e(2,3), c02 = e(1,2) * e(3,3) - e(3,2) * e(1,3), c03 = e(1,2) * e(2,3) - e(2,2) * e(1,3);
    const float c04 = e(2,1) * e(3,3) - e(3,1) * e(2,3), c06 = e(1,1) * e(3,3) - e(3,1) * e(1,3), c07 = e(1,1) * e(2,3) - e(2,1) * e(1,3);
    const float c08 = e(2,1) * e(3,2) - e(3,1) * e(2,2), c10 = e(1,1) * e(3,2) - e(3,1) * e(1,2), c11 = e(1,1) * e(2,2) - e(2,1) * e(1,2);
    const float c12 = e(2,0) * e(3,3) - e(3,0) * e(2,3), c14 = e(1,0) * e(3,3) - e(3,0) * e(1,3), c15 = e(1,0) * e(2,3) - e(2,0) * e(1,3);
    const float c16 = e(2,0) * e(3,2) - e(3,0) * e(2,2), c18 = e(1,0) * e(3,2) - e(3,0) * e(1,2), c19 = e(1,0) * e(2,2) - e(2,0) * e(1,2);
    const float c20 = e(2,0) * e(3,1) - e(3,0) * e(2,1), c22 = e(1,0) * e(3,1) - e(3,0) * e(1,1), c23 = e(1,0) * e(2,1) - e(2,0) * e(1,1);
    const float F0[4] = {c00, c00, c02, c03}, F1[4] = {c04, c04, c06, c07}, F2[4] = {c08, c08, c10, c11};
    const float F3[4] = {c12, c12, c14, c15}, F4[4] = {c16, c16, c18, c19}, F5[4] = {c20, c20, c22, c23};
    const float V0[4] = {e(1,0), e(0,0), e(0,0), e(0,0)}, V1[4] = {e(1,1), e(0,1), e(0,1), e(0,1)};
    const float V2[4] = {e(1,2), e(0,2), e(0,2), e(0,2)}, V3[4] = {e(1,3), e(0,3), e(0,3), e(0,3)};
    const float SA[4] = {+1, -1, +1, -1}, SB[4] = {-1, +1, -1, +1};
    Mat4 I;
    for (int k = 0; k < 4; k++) {
        I.m[0 + k] = ((V1[k] * F0[k] - V2[k] * F1[k]) + V3[k] * F2[k]) * SA[k];
        I.m[4 + k] = ((V0[k] * F0[k] - V2[k] * F3[k]) + V3[k] * F4[k]) * SB[k];
        I.m[8 + k] = ((V0[k] * F1[k] - V1[k] * F3[k]) + V3[k] * F5[k]) * SA[k];
        I.m[12 + k] = ((V0[k] * F2[k] - V1[k] * F4[k]) + V2[k] * F5[k]) * SB[k];
    }
    const float d0 = e(0,0) * I.m[0], d1 = e(0,1) * I.m[4], d2 = e(0,2) * I.m[8], d3 = e(0,3) * I.m[12];
    const float oneOverDet = 1.0f / ((d0 + d1) + (d2 + d3));
    for (int k = 0; k < 16; k++) I.m[k] = I.m[k] * oneOverDet;
    return I;
}

// glm::inverseTranspose(mat4) (gtc/matrix_inverse.inl:93-147), SubFactor11 as glm 0.9.6.3 has it
inline Mat4 inverseTranspose(const Mat4 &A) {
    auto e = [&](int c, int r) { return A.m[c * 4 + r]; };
    const float s00 = e(2,2) * e(3,3) - e(3,2) * e(2,3), s01 = e(2,1) * e(3,3) - e(3,1) * e(2,3), s02 = e(2,1) * e(3,2) - e(3,1) * e(2,2);
    const float s03 = e(2,0) * e(3,3) - e(3,0) * e(2,3), s04 = e(2,0) * e(3,2) - e(3,0) * e(2,2), s05 = e(2,0) * e(3,1) - e(3,0) * e(2,1);
    const float s06 = e(1,2) * e(3,3) - e(3,2) * e(1,3), s07 = e(1,1) * e(3,3) - e(3,1) * e(1,3), s08 = e(1,1) * e(3,2) - e(3,1) * e(1,2);
    const float s09 = e(1,0) * e(3,3) - e(3,0) * e(1,3), s10 = e(1,0) * e(3,2) - e(3,0) * e(1,2), s11 = e(1,1) * e(3,3) - e(3,1) * e(1,3);
    const float s12 = e(1,0) * e(3,1) - e(3,0) * e(1,1), s13 = e(1,2) * e(2,3) - e(2,2) * e(1,3), s14 = e(1,1) * e(2,3) - e(2,1) * e(1,3);
    const float s15 = e(1,1) * e(2,2) - e(2,1) * e(1,2), s16 = e(1,0) * e(2,3) - e(2,0) * e(1,3), s17 = e(1,0) * e(2,2) - e(2,0) * e(1,2);
    const float s18 = e(1,0) * e(2,1) - e(2,0) * e(1,1);
    Mat4 I;
    I.m[0] = +((e(1,1) * s00 - e(1,2) * s01) + e(1,3) * s02);
    I.m[1] = -((e(1,0) * s00 - e(1,2) * s03) + e(1,3) * s04);
    I.m[2] = +((e(1,0) * s01 - e(1,1) * s03) + e(1,3) * s05);
    I.m[3] = -((e(1,0) * s02 - e(1,1) * s04) + e(1,2) * s05);
    I.m[4] = -((e(0,1) * s00 - e(0,2) * s01) + e(0,3) * s02);
    I.m[5] = +((e(0,0) * s00 - e(0,2) * s03) + e(0,3) * s04);
    I.m[6] = -((e(0,0) * s01 - e(0,1) * s03) + e(0,3) * s05);
    I.m[7] = +((e(0,0) * s02 - e(0,1) * s04) + e(0,2) * s05);
    I.m[8] = +((e(0,1) * s06 - e(0,2) * s07) + e(0,3) * s08);
    I.m[9] = -((e(0,0) * s06 - e(0,2) * s09) + e(0,3) * s10);
    I.m[10] = +((e(0,0) * s11 - e(0,1) * s09) + e(0,3) * s12);
    I.m[11] = -((e(0,0) * s08 - e(0,1) * s10) + e(0,2) * s12);
    I.m[12] = -((e(0,1) * s13 - e(0,2) * s14) + e(0,3) * s15);
    I.m[13] = +((e(0,0) * s13 - e(0,2) * s16) + e(0,3) * s17);
    I.m[14] = -((e(0,0) * s14 - e(0,1) * s16) + e(0,3) * s18);
    I.m[15] = +((e(0,0) * s15 - e(0,1) * s17) + e(0,2) * s18);
    const float det = ((+e(0,0) * I.m[0] + e(0,1) * I.m[1]) + e(0,2) * I.m[2]) + e(0,3) * I.m[3];
    for (int k = 0; k < 16; k++) I.m[k] = I.m[k] / det;
    return I;
}

#define PTH_PI 3.1415926535897932384626422832795028841971f      // src/utilities.h:12

// utilityCore::buildTransformationMatrix (src/utilities.cpp:65-72)
inline Mat4 buildTransformationMatrix(const float t[3], const float rot[3], const float s[3]) {
    const float X[3] = {1, 0, 0}, Y[3] = {0, 1, 0}, Z[3] = {0, 0, 1};
    const Mat4 I = identity();
    Mat4 T = translate(I, t);
    Mat4 R = rotate(I, rot[0] * PTH_PI / 180, X);
    R = mul(R, rotate(I, rot[1] * PTH_PI / 180, Y));
    R = mul(R, rotate(I, rot[2] * PTH_PI / 180, Z));
    Mat4 S = scale(I, s);
    return mul(mul(T, R), S);
}

inline void norm3(const float v[3], float out[3]) {
    const float inv = 1.0f / sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    out[0] = v[0] * inv; out[1] = v[1] * inv; out[2] = v[2] * inv;
}
inline void cross3(const float x[3], const float y[3], float out[3]) {
    const float a = x[1] * y[2] - y[1] * x[2], b = x[2] * y[0] - y[2] * x[0], c = x[0] * y[1] - y[0] * x[1];
    out[0] = a; out[1] = b; out[2] = c;
}

}  // namespace pth
