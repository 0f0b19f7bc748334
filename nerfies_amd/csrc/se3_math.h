// Closed forms of the SE(3) exponential applied to a point, its vector-Jacobian product, and the Dual number they are evaluated
// on for the warp Jacobian / its Hessian-vector products (rigid_body.py:21-97, warping.py:330-344).  Shared by the
// fp32 (warp_chain.hip) and bf16 (warp_bf16.hip) SE3 kernels: the trunk may run on bf16 operands, this algebra is always fp32.
#pragma once
#include <hip/hip_runtime.h>

namespace nrf {

// Scalar with one forward-mode tangent: the warp Jacobian (jax.jacfwd(self.warp), warping.py:385-387) and the
// Hessian-vector products its reverse pass needs both come from running the SAME closed forms on Duals.
struct Dual {
  float v, d;
  __device__ __forceinline__ Dual() : v(0.f), d(0.f) {}
  __device__ __forceinline__ Dual(float a) : v(a), d(0.f) {}
  __device__ __forceinline__ Dual(float a, float b) : v(a), d(b) {}
};
__device__ __forceinline__ Dual operator+(Dual a, Dual b) { return Dual(a.v + b.v, a.d + b.d); }
__device__ __forceinline__ Dual operator-(Dual a, Dual b) { return Dual(a.v - b.v, a.d - b.d); }
__device__ __forceinline__ Dual operator*(Dual a, Dual b) { return Dual(a.v * b.v, a.v * b.d + a.d * b.v); }
__device__ __forceinline__ Dual operator/(Dual a, Dual b) { const float q = a.v / b.v; return Dual(q, (a.d - q * b.d) / b.v); }
__device__ __forceinline__ float val(float a) { return a; }
__device__ __forceinline__ float val(Dual a) { return a.v; }
__device__ __forceinline__ float sqrt_t(float a) { return sqrtf(a); }
__device__ __forceinline__ Dual sqrt_t(Dual a) { const float r = sqrtf(a.v); return Dual(r, 0.5f * a.d / r); }
__device__ __forceinline__ void sincos_t(float a, float& s, float& c) { sincosf(a, &s, &c); }
__device__ __forceinline__ void sincos_t(Dual a, Dual& s, Dual& c) {
  float sv, cv;
  sincosf(a.v, &sv, &cv);
  s = Dual(sv, cv * a.d); c = Dual(cv, -sv * a.d);
}

template <typename T> struct V3T { T x, y, z; };
typedef V3T<float> V3;
template <typename T> __device__ __forceinline__ V3T<T> v3t(T x, T y, T z) { V3T<T> r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 v3(float x, float y, float z) { return v3t<float>(x, y, z); }
template <typename T> __device__ __forceinline__ V3T<T> operator+(V3T<T> a, V3T<T> b) { return v3t<T>(a.x + b.x, a.y + b.y, a.z + b.z); }
template <typename T> __device__ __forceinline__ V3T<T> operator-(V3T<T> a, V3T<T> b) { return v3t<T>(a.x - b.x, a.y - b.y, a.z - b.z); }
template <typename T> __device__ __forceinline__ V3T<T> operator*(T s, V3T<T> a) { return v3t<T>(s * a.x, s * a.y, s * a.z); }
template <typename T> __device__ __forceinline__ T dot(V3T<T> a, V3T<T> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <typename T> __device__ __forceinline__ V3T<T> cross(V3T<T> a, V3T<T> b) {
  return v3t<T>(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

// Coefficients of the closed form of exp_se3 applied to a point, as functions of
// t2 = |w|^2:  A = sin t / t,  B = (1 - cos t) / t^2,  C = (t - sin t) / t^3  and their
// derivatives  dA/dw = Ab w,  dB/dw = Bb w,  dC/dw = Cb w  with
//   Ab = C - B (= (t cos t - sin t)/t^3),  Bb = (A - 2B)/t^2,  Cb = (B - 3C)/t^2.
// The reference evaluates the un-simplified normalised-axis form in fp32 (rigid_body.py:54-89),
// whose 1-cos / t-sin terms cancel catastrophically for small angles and are NaN at t = 0; here
// small angles use the Taylor series, so the result tracks the exact value to fp32 rounding.
template <typename T> struct Se3Coef { T A, B, C, Ab, Bb, Cb; };
template <typename T>
__device__ __forceinline__ Se3Coef<T> se3_coef(T t2) {
  Se3Coef<T> c;
  if (val(t2) < 0.04f) {
    c.A = T(1.f) + t2 * (T(-1.f / 6.f) + t2 * (T(1.f / 120.f) + t2 * T(-1.f / 5040.f)));
    c.B = T(0.5f) + t2 * (T(-1.f / 24.f) + t2 * (T(1.f / 720.f) + t2 * T(-1.f / 40320.f)));
    c.C = T(1.f / 6.f) + t2 * (T(-1.f / 120.f) + t2 * (T(1.f / 5040.f) + t2 * T(-1.f / 362880.f)));
    c.Ab = T(-1.f / 3.f) + t2 * (T(1.f / 30.f) + t2 * (T(-1.f / 840.f) + t2 * T(1.f / 45360.f)));
    c.Bb = T(-1.f / 12.f) + t2 * (T(1.f / 180.f) + t2 * (T(-1.f / 6720.f) + t2 * T(1.f / 453600.f)));
    c.Cb = T(-1.f / 60.f) + t2 * (T(1.f / 1260.f) + t2 * (T(-1.f / 60480.f) + t2 * T(1.f / 4989600.f)));
  } else {
    const T t = sqrt_t(t2);
    T s, co, sh, ch;
    sincos_t(t, s, co);
    sincos_t(T(0.5f) * t, sh, ch);
    c.A = s / t;
    c.B = T(2.f) * sh * sh / t2;
    c.C = (t - s) / (t2 * t);
    c.Ab = c.C - c.B;
    c.Bb = (c.A - T(2.f) * c.B) / t2;
    c.Cb = (c.B - T(3.f) * c.C) / t2;
  }
  return c;
}

// The same coefficients on Duals that share ONE value part (the warp Jacobian / its Hessian-vector products evaluate the closed
// forms along three directions at the same (w, v, x): t2 = w.w has the same value and three tangents).  Every coefficient is a
// function of t2 alone, so  X(Dual(t2, d)) = Dual(X, X'(t2) d):  values and t2-derivatives are formed ONCE (Se3CoefD) and each
// direction costs six multiplies instead of a second evaluation of the divisions / sincos (round 6: the elastic kernel spent most
// of its 6 k instructions re-deriving them six times).  From dX/dw = Xb w and dt2/dw = 2 w:
//   A' = Ab / 2,  B' = Bb / 2,  C' = Cb / 2,  Ab' = (Cb - Bb) / 2,  Bb' = (Ab / 2 - 2 Bb) / t2,  Cb' = (Bb / 2 - 5 Cb / 2) / t2;
// below t2 = 0.04 Bb' and Cb' are the derivatives of the Taylor polynomials above (no division by a small t2).
struct Se3CoefD { Se3Coef<float> c; float dAb, dBb, dCb; };
__device__ __forceinline__ Se3CoefD se3_coef_d(float t2) {
  Se3CoefD r;
  r.c = se3_coef<float>(t2);
  r.dAb = 0.5f * (r.c.Cb - r.c.Bb);
  if (t2 < 0.04f) {
    r.dBb = 1.f / 180.f + t2 * (-1.f / 3360.f + t2 * (1.f / 151200.f));
    r.dCb = 1.f / 1260.f + t2 * (-1.f / 30240.f + t2 * (1.f / 1663200.f));
  } else {
    const float it2 = 1.f / t2;
    r.dBb = (0.5f * r.c.Ab - 2.f * r.c.Bb) * it2;
    r.dCb = (0.5f * r.c.Bb - 2.5f * r.c.Cb) * it2;
  }
  return r;
}
// the Dual coefficients along a direction whose t2 tangent is t2d (= 2 w . wd)
__device__ __forceinline__ Se3Coef<Dual> se3_coef_along(const Se3CoefD& k, float t2d) {
  Se3Coef<Dual> c;
  c.A = Dual(k.c.A, 0.5f * k.c.Ab * t2d); c.B = Dual(k.c.B, 0.5f * k.c.Bb * t2d); c.C = Dual(k.c.C, 0.5f * k.c.Cb * t2d);
  c.Ab = Dual(k.c.Ab, k.dAb * t2d); c.Bb = Dual(k.c.Bb, k.dBb * t2d); c.Cb = Dual(k.c.Cb, k.dCb * t2d);
  return c;
}

// x' - x = exp_se3([w; v]) x - x = A w*x + B w*(w*x) + v + B w*v + C w*(w*v)   (warping.py:330-344)
template <typename T>
__device__ __forceinline__ V3T<T> se3_delta_c(const Se3Coef<T>& c, V3T<T> w, V3T<T> v, V3T<T> x) {
  const V3T<T> wx = cross(w, x), wv = cross(w, v);
  const V3T<T> wwx = cross(w, wx), wwv = cross(w, wv);
  return c.A * wx + c.B * wwx + v + c.B * wv + c.C * wwv;
}
template <typename T>
__device__ __forceinline__ V3T<T> se3_delta(V3T<T> w, V3T<T> v, V3T<T> x) {
  return se3_delta_c<T>(se3_coef<T>(dot(w, w)), w, v, x);
}
__device__ __forceinline__ V3 se3_apply(V3 w, V3 v, V3 x) { return x + se3_delta<float>(w, v, x); }

// VJP of se3_apply for upstream g = dL/dx':  dL/dw, dL/dv  (dL/dx is not needed: sample points
// carry no parameters).  On Duals the value parts are the VJP, the tangent parts its directional
// derivative = the Hessian-vector product of g . exp_se3(w, v) x along the Dual direction.
template <typename T>
__device__ __forceinline__ void se3_vjp_c(const Se3Coef<T>& c, V3T<T> w, V3T<T> v, V3T<T> x, V3T<T> g, V3T<T>& dw, V3T<T>& dv) {
  const V3T<T> gw = cross(g, w);            // g x w
  const V3T<T> wgw = cross(w, cross(w, g)); // w x (w x g)
  dv = g + c.B * gw + c.C * wgw;            // V^T g
  const V3T<T> wx = cross(w, x), wv = cross(w, v);
  const V3T<T> wwx = cross(w, wx), wwv = cross(w, wv);
  const T wg = dot(w, g);
  auto D = [&](V3T<T> y) { return wg * y + dot(w, y) * g - (T(2.f) * dot(y, g)) * w; };
  const T sa = c.Ab * dot(g, wx) + c.Bb * (dot(g, wwx) + dot(g, wv)) + c.Cb * dot(g, wwv);
  dw = c.A * cross(x, g) + c.B * (D(x) + cross(v, g)) + c.C * D(v) + sa * w;
}
template <typename T>
__device__ __forceinline__ void se3_vjp(V3T<T> w, V3T<T> v, V3T<T> x, V3T<T> g, V3T<T>& dw, V3T<T>& dv) {
  se3_vjp_c<T>(se3_coef<T>(dot(w, w)), w, v, x, g, dw, dv);
}

}  // namespace nrf
