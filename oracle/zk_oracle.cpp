// oracle/zk_oracle.cpp -- TEST INFRASTRUCTURE ONLY.
//
// CPU restatement (64-bit limbs, unsigned __int128) of the algorithms the zksnake hot path
// executes inside arkworks 0.4.x (not vendored in /root/reference, not buildable here: no Rust).
// Used ONLY by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker
// and as the timed "port" CPU baseline.  The product library (zksnake_amd/csrc) never links,
// loads or calls anything in this file; it uses a different limb width (32-bit) and a
// different code base, so agreement between the two is an independent check.
//
// PARITY UNPINNED against the reference binary (see oracle/pyref.py header and DESIGN.md);
// this file is itself pinned against oracle/pyref.py (definitions + public KATs) by
// tests/test_oracle.py.
//
// Restated call sites (file:line under /root/reference):
//   multiscalar_mul_g1/g2 -> src/bn254/curve.rs:356-392, src/bls12_381/curve.rs:366-402
//        = ark-ec 0.4.2 VariableBaseMSM::msm -> msm_bigint_wnaf: signed-digit Pippenger,
//          window c = (n < 32) ? 3 : floor(log2(n)*69/100) + 2, 2^(c-1) buckets per window,
//          running-sum bucket reduction, Horner over windows.
//   fft/ifft              -> src/bn254/polynomial.rs:535-571 = ark-poly 0.4.2 Radix2EvaluationDomain
//          (serial radix-2, natural order in/out, inverse scales by 1/N).
//   mul_over_evaluation_domain -> src/bn254/polynomial.rs:609-634
//   divide_by_vanishing_poly   -> src/bn254/polynomial.rs:466-489
//   batch_multi_scalar_g1/g2   -> src/bn254/curve.rs:326-354 (independent double-and-add)
//   PointG1/G2 __add__/__mul__ -> src/bn254/curve.rs:77-106,255-284
//
// ABI of this file: canonical (non-Montgomery) integers as little-endian 64-bit limbs;
// affine points (x, y) with (0,0) = point at infinity; Fp2 as (c0, c1).

#include <cstdint>
#include <cstring>
#include <cstdlib>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;

// ------------------------------------------------------------------------------------------
// prime fields
// ------------------------------------------------------------------------------------------

template <int N>
struct FpCtx {
    uint64_t p[N];
    uint64_t r2[N];   // R^2 mod p
    uint64_t one[N];  // R mod p
    uint64_t inv;     // -p^-1 mod 2^64
    int bits;
};

template <int N>
static inline bool ge(const uint64_t* a, const uint64_t* b) {
    for (int i = N - 1; i >= 0; --i) {
        if (a[i] != b[i]) return a[i] > b[i];
    }
    return true;
}

template <int N>
static inline uint64_t sub_n(uint64_t* r, const uint64_t* a, const uint64_t* b) {
    uint64_t borrow = 0;
    for (int i = 0; i < N; ++i) {
        u128 t = (u128)a[i] - b[i] - borrow;
        r[i] = (uint64_t)t;
        borrow = (uint64_t)(t >> 64) & 1;
    }
    return borrow;
}

template <int N>
static inline uint64_t add_n(uint64_t* r, const uint64_t* a, const uint64_t* b) {
    uint64_t carry = 0;
    for (int i = 0; i < N; ++i) {
        u128 t = (u128)a[i] + b[i] + carry;
        r[i] = (uint64_t)t;
        carry = (uint64_t)(t >> 64);
    }
    return carry;
}

// build a context from the modulus alone (R^2 by repeated doubling; inv by Newton iteration)
template <int N>
static FpCtx<N> make_ctx(const uint64_t* p) {
    FpCtx<N> c;
    memcpy(c.p, p, sizeof(c.p));
    uint64_t inv = 1;
    for (int i = 0; i < 6; ++i) inv *= 2 - p[0] * inv;
    c.inv = (uint64_t)0 - inv;
    // x = 1; double 2*64*N times mod p -> R^2; after 64*N doublings -> R
    uint64_t x[N];
    memset(x, 0, sizeof(x));
    x[0] = 1;
    for (int i = 0; i < 2 * 64 * N; ++i) {
        uint64_t carry = add_n<N>(x, x, x);
        if (carry || ge<N>(x, p)) sub_n<N>(x, x, p);
        if (i == 64 * N - 1) memcpy(c.one, x, sizeof(x));
    }
    memcpy(c.r2, x, sizeof(x));
    int bits = 64 * N;
    while (bits > 0 && !((p[(bits - 1) / 64] >> ((bits - 1) % 64)) & 1)) --bits;
    c.bits = bits;
    return c;
}

template <class Tag>
struct Fp {
    static constexpr int N = Tag::N;
    uint64_t v[Tag::N];

    static const FpCtx<Tag::N>& C() { return Tag::ctx(); }

    static Fp zero() { Fp r; memset(r.v, 0, sizeof(r.v)); return r; }
    static Fp one() { Fp r; memcpy(r.v, C().one, sizeof(r.v)); return r; }
    bool is_zero() const { uint64_t o = 0; for (int i = 0; i < N; ++i) o |= v[i]; return o == 0; }
    bool operator==(const Fp& o) const { return memcmp(v, o.v, sizeof(v)) == 0; }
    bool operator!=(const Fp& o) const { return !(*this == o); }

    Fp operator+(const Fp& o) const {
        Fp r;
        uint64_t carry = add_n<N>(r.v, v, o.v);
        if (carry || ge<N>(r.v, C().p)) sub_n<N>(r.v, r.v, C().p);
        return r;
    }
    Fp operator-(const Fp& o) const {
        Fp r;
        if (sub_n<N>(r.v, v, o.v)) add_n<N>(r.v, r.v, C().p);
        return r;
    }
    Fp neg() const {
        if (is_zero()) return *this;
        Fp r; sub_n<N>(r.v, C().p, v); return r;
    }
    Fp dbl() const { return *this + *this; }

    // Montgomery CIOS
    Fp operator*(const Fp& o) const {
        const uint64_t* p = C().p;
        const uint64_t inv = C().inv;
        uint64_t t[N + 2];
        memset(t, 0, sizeof(t));
        for (int i = 0; i < N; ++i) {
            uint64_t carry = 0;
            for (int j = 0; j < N; ++j) {
                u128 s = (u128)v[j] * o.v[i] + t[j] + carry;
                t[j] = (uint64_t)s;
                carry = (uint64_t)(s >> 64);
            }
            u128 s = (u128)t[N] + carry;
            t[N] = (uint64_t)s;
            t[N + 1] = (uint64_t)(s >> 64);
            uint64_t m = t[0] * inv;
            u128 s2 = (u128)m * p[0] + t[0];
            carry = (uint64_t)(s2 >> 64);
            for (int j = 1; j < N; ++j) {
                s2 = (u128)m * p[j] + t[j] + carry;
                t[j - 1] = (uint64_t)s2;
                carry = (uint64_t)(s2 >> 64);
            }
            s2 = (u128)t[N] + carry;
            t[N - 1] = (uint64_t)s2;
            t[N] = t[N + 1] + (uint64_t)(s2 >> 64);
        }
        Fp r;
        if (t[N] || ge<N>(t, p)) sub_n<N>(r.v, t, p);
        else memcpy(r.v, t, sizeof(r.v));
        return r;
    }
    Fp sqr() const { return *this * *this; }

    static Fp from_canonical(const uint64_t* a) {
        Fp x; memcpy(x.v, a, sizeof(x.v));
        // reduce (inputs may be >= p): subtract p while needed (inputs < 2^(64N))
        while (ge<N>(x.v, C().p)) sub_n<N>(x.v, x.v, C().p);
        Fp r2; memcpy(r2.v, C().r2, sizeof(r2.v));
        return x * r2;
    }
    void to_canonical(uint64_t* out) const {
        Fp o; memset(o.v, 0, sizeof(o.v)); o.v[0] = 1;
        Fp r = *this * o;
        memcpy(out, r.v, sizeof(r.v));
    }
    static Fp from_u64(uint64_t x) {
        uint64_t a[N]; memset(a, 0, sizeof(a)); a[0] = x; return from_canonical(a);
    }
    Fp pow(const uint64_t* e, int nlimbs) const {
        Fp acc = one();
        for (int i = nlimbs * 64 - 1; i >= 0; --i) {
            acc = acc.sqr();
            if ((e[i / 64] >> (i % 64)) & 1) acc = acc * *this;
        }
        return acc;
    }
    Fp inv() const {
        uint64_t e[N]; uint64_t two[N]; memset(two, 0, sizeof(two)); two[0] = 2;
        sub_n<N>(e, C().p, two);
        return pow(e, N);
    }
};

template <class Tag>
struct Fp2 {
    typedef Fp<Tag> B;
    B c0, c1;
    static Fp2 zero() { return {B::zero(), B::zero()}; }
    static Fp2 one() { return {B::one(), B::zero()}; }
    bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
    bool operator==(const Fp2& o) const { return c0 == o.c0 && c1 == o.c1; }
    bool operator!=(const Fp2& o) const { return !(*this == o); }
    Fp2 operator+(const Fp2& o) const { return {c0 + o.c0, c1 + o.c1}; }
    Fp2 operator-(const Fp2& o) const { return {c0 - o.c0, c1 - o.c1}; }
    Fp2 neg() const { return {c0.neg(), c1.neg()}; }
    Fp2 dbl() const { return {c0.dbl(), c1.dbl()}; }
    Fp2 operator*(const Fp2& o) const {
        B a = c0 * o.c0, b = c1 * o.c1;
        B c = (c0 + c1) * (o.c0 + o.c1);
        return {a - b, c - a - b};
    }
    Fp2 sqr() const {
        B a = (c0 + c1) * (c0 - c1);
        B b = c0 * c1;
        return {a, b.dbl()};
    }
    Fp2 inv() const {
        B d = (c0.sqr() + c1.sqr()).inv();
        return {c0 * d, (c1 * d).neg()};
    }
    static constexpr int WORDS = 2 * Tag::N;
    static Fp2 from_canonical(const uint64_t* a) {
        return {B::from_canonical(a), B::from_canonical(a + Tag::N)};
    }
    void to_canonical(uint64_t* out) const { c0.to_canonical(out); c1.to_canonical(out + Tag::N); }
};

template <class Tag>
struct Fp1 : Fp<Tag> {  // adds the WORDS constant so curves can be generic over Fp / Fp2
    static constexpr int WORDS = Tag::N;
    Fp1() {}
    Fp1(const Fp<Tag>& o) : Fp<Tag>(o) {}
};

// ------------------------------------------------------------------------------------------
// field instances
// ------------------------------------------------------------------------------------------

static const uint64_t BN_Q[4] = {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
static const uint64_t BN_R[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
static const uint64_t BLS_Q[6] = {0xb9feffffffffaaabULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL,
                                  0x64774b84f38512bfULL, 0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL};
static const uint64_t BLS_R[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL};

struct BnFqTag { static constexpr int N = 4; static const FpCtx<4>& ctx() { static FpCtx<4> c = make_ctx<4>(BN_Q); return c; } };
struct BnFrTag { static constexpr int N = 4; static const FpCtx<4>& ctx() { static FpCtx<4> c = make_ctx<4>(BN_R); return c; } };
struct BlsFqTag { static constexpr int N = 6; static const FpCtx<6>& ctx() { static FpCtx<6> c = make_ctx<6>(BLS_Q); return c; } };
struct BlsFrTag { static constexpr int N = 4; static const FpCtx<4>& ctx() { static FpCtx<4> c = make_ctx<4>(BLS_R); return c; } };

// ------------------------------------------------------------------------------------------
// short Weierstrass, a = 0, Jacobian coordinates; Z == 0 is the point at infinity
// ------------------------------------------------------------------------------------------

template <class K>
struct Aff { K x, y; bool inf; };

template <class K>
struct Jac {
    K X, Y, Z;
    static Jac infinity() { return {K::one(), K::one(), K::zero()}; }
    bool is_inf() const { return Z.is_zero(); }

    Jac dbl() const {
        if (is_inf() || Y.is_zero()) return infinity();
        K A = X.sqr(), B = Y.sqr(), C = B.sqr();
        K t = (X + B).sqr() - A - C;
        K D = t.dbl();
        K E = A.dbl() + A;
        K F = E.sqr();
        K X3 = F - D.dbl();
        K C8 = C.dbl().dbl().dbl();
        K Y3 = E * (D - X3) - C8;
        K Z3 = (Y * Z).dbl();
        return {X3, Y3, Z3};
    }
    Jac add_affine(const Aff<K>& q) const {
        if (q.inf) return *this;
        if (is_inf()) return {q.x, q.y, K::one()};
        K Z1Z1 = Z.sqr();
        K U2 = q.x * Z1Z1;
        K S2 = q.y * Z * Z1Z1;
        if (U2 == X) {
            if (S2 == Y) return dbl();
            return infinity();
        }
        K H = U2 - X, HH = H.sqr(), HHH = H * HH;
        K r = S2 - Y;
        K V = X * HH;
        K X3 = r.sqr() - HHH - V.dbl();
        K Y3 = r * (V - X3) - Y * HHH;
        K Z3 = Z * H;
        return {X3, Y3, Z3};
    }
    Jac add(const Jac& q) const {
        if (q.is_inf()) return *this;
        if (is_inf()) return q;
        K Z1Z1 = Z.sqr(), Z2Z2 = q.Z.sqr();
        K U1 = X * Z2Z2, U2 = q.X * Z1Z1;
        K S1 = Y * q.Z * Z2Z2, S2 = q.Y * Z * Z1Z1;
        if (U1 == U2) {
            if (S1 == S2) return dbl();
            return infinity();
        }
        K H = U2 - U1, HH = H.sqr(), HHH = H * HH;
        K r = S2 - S1;
        K V = U1 * HH;
        K X3 = r.sqr() - HHH - V.dbl();
        K Y3 = r * (V - X3) - S1 * HHH;
        K Z3 = Z * q.Z * H;
        return {X3, Y3, Z3};
    }
    Jac neg() const { return {X, Y.neg(), Z}; }
    Aff<K> to_affine() const {
        if (is_inf()) return {K::zero(), K::zero(), true};
        K zi = Z.inv(), zi2 = zi.sqr();
        return {X * zi2, Y * zi2 * zi, false};
    }
};

template <class K>
static Aff<K> load_aff(const uint64_t* src) {
    bool z = true;
    for (int i = 0; i < 2 * K::WORDS; ++i) z = z && (src[i] == 0);
    if (z) return {K::zero(), K::zero(), true};
    return {K::from_canonical(src), K::from_canonical(src + K::WORDS), false};
}

template <class K>
static void store_aff(uint64_t* dst, const Aff<K>& a) {
    if (a.inf) { memset(dst, 0, sizeof(uint64_t) * 2 * K::WORDS); return; }
    a.x.to_canonical(dst);
    a.y.to_canonical(dst + K::WORDS);
}

// scalars are 4x64 canonical; reduce mod r first (Fr::from(BigUint) semantics)
template <class FrTag>
static void reduce_scalar(uint64_t* out, const uint64_t* in) {
    memcpy(out, in, 32);
    const uint64_t* r = FrTag::ctx().p;
    while (ge<4>(out, r)) sub_n<4>(out, out, r);
}

template <class K>
static Jac<K> scalar_mul(const Aff<K>& base, const uint64_t* k) {
    Jac<K> acc = Jac<K>::infinity();
    bool started = false;
    for (int i = 255; i >= 0; --i) {
        if (started) acc = acc.dbl();
        if ((k[i / 64] >> (i % 64)) & 1) { acc = acc.add_affine(base); started = true; }
    }
    return acc;
}

// ark-ec 0.4.2 msm_bigint_wnaf restated
static int ark_window(size_t n) {
    if (n < 32) return 3;
    int lg = 63 - __builtin_clzll((unsigned long long)n);
    // ark_std::log2(n) is ceil(log2); ln_without_floats(a) = log2(a) * 69 / 100
    int lg_ceil = ((size_t)1 << lg) == n ? lg : lg + 1;
    return lg_ceil * 69 / 100 + 2;
}

// Window count: ark uses ceil(num_bits / w) and folds the last carry back into the top digit; when
// w divides num_bits (BLS12-381 Fr is 255 bits: w = 3, 5, 15, 17) that top digit can exceed the
// 2^(w-1) buckets, so this restatement always keeps one spare bit: floor(num_bits / w) + 1
// windows (identical to ark's count whenever w does not divide num_bits), final carry == 0.
static int window_count(int num_bits, int w) { return num_bits / w + 1; }

static void make_digits(const uint64_t* a, int w, int num_bits, std::vector<int64_t>& out, size_t off, size_t stride) {
    const uint64_t radix = 1ULL << w;
    const uint64_t mask = radix - 1;
    uint64_t carry = 0;
    int count = window_count(num_bits, w);
    for (int i = 0; i < count; ++i) {
        int bit_offset = i * w;
        int u = bit_offset / 64, b = bit_offset % 64;
        uint64_t bits;
        if (u > 3) bits = 0;
        else if (b < 64 - w || u == 3) bits = a[u] >> b;
        else bits = (a[u] >> b) | (a[u + 1] << (64 - b));
        uint64_t coef = carry + (bits & mask);
        carry = (coef + radix / 2) >> w;
        int64_t d = (int64_t)coef - (int64_t)(carry << w);
        out[off + (size_t)i * stride] = d;
    }
}

template <class K, class FrTag>
static Jac<K> pippenger(size_t n, const uint64_t* scalars, const uint64_t* bases, int threads, int c_override) {
    if (n == 0) return Jac<K>::infinity();
    const int num_bits = FrTag::ctx().bits;
    const int c = c_override > 0 ? c_override : ark_window(n);
    const int nwin = window_count(num_bits, c);
    std::vector<Aff<K>> pts(n);
    for (size_t i = 0; i < n; ++i) pts[i] = load_aff<K>(bases + i * 2 * K::WORDS);
    std::vector<int64_t> digits((size_t)nwin * n);
    for (size_t i = 0; i < n; ++i) {
        uint64_t s[4];
        reduce_scalar<FrTag>(s, scalars + 4 * i);
        make_digits(s, c, num_bits, digits, i, n);
    }
    std::vector<Jac<K>> wsum(nwin);
    (void)threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : 1)
#endif
    for (int w = 0; w < nwin; ++w) {
        std::vector<Jac<K>> buckets((size_t)1 << (c - 1), Jac<K>::infinity());
        const int64_t* d = &digits[(size_t)w * n];
        for (size_t i = 0; i < n; ++i) {
            int64_t dg = d[i];
            if (dg > 0) buckets[dg - 1] = buckets[dg - 1].add_affine(pts[i]);
            else if (dg < 0) {
                Aff<K> m = pts[i];
                m.y = m.y.neg();
                buckets[-dg - 1] = buckets[-dg - 1].add_affine(m);
            }
        }
        Jac<K> running = Jac<K>::infinity(), res = Jac<K>::infinity();
        for (size_t b = buckets.size(); b-- > 0;) {
            running = running.add(buckets[b]);
            res = res.add(running);
        }
        wsum[w] = res;
    }
    Jac<K> total = wsum[nwin - 1];
    for (int w = nwin - 2; w >= 0; --w) {
        for (int i = 0; i < c; ++i) total = total.dbl();
        total = total.add(wsum[w]);
    }
    return total;
}

// ------------------------------------------------------------------------------------------
// NTT (serial radix-2; natural order in and out)
// ------------------------------------------------------------------------------------------

template <class FrTag>
static Fp<FrTag> root_of_unity(int log_n, int two_adicity, uint64_t gen) {
    typedef Fp<FrTag> F;
    // g^((r-1)/2^s)
    uint64_t e[4];
    uint64_t one[4] = {1, 0, 0, 0};
    sub_n<4>(e, FrTag::ctx().p, one);
    // shift right by two_adicity
    for (int k = 0; k < two_adicity; ++k) {
        for (int i = 0; i < 3; ++i) e[i] = (e[i] >> 1) | (e[i + 1] << 63);
        e[3] >>= 1;
    }
    F w = F::from_u64(gen).pow(e, 4);
    for (int k = 0; k < two_adicity - log_n; ++k) w = w.sqr();
    return w;
}

template <class FrTag>
static int ntt_inplace(uint64_t* data, int log_n, int inverse, int two_adicity, uint64_t gen, int threads) {
    typedef Fp<FrTag> F;
    if (log_n > two_adicity) return 2;
    const size_t n = (size_t)1 << log_n;
    std::vector<F> a(n);
    for (size_t i = 0; i < n; ++i) a[i] = F::from_canonical(data + 4 * i);
    F w = root_of_unity<FrTag>(log_n, two_adicity, gen);
    if (inverse) w = w.inv();
    for (size_t i = 0; i < n; ++i) {
        size_t j = 0;
        for (int b = 0; b < log_n; ++b) j |= ((i >> b) & 1) << (log_n - 1 - b);
        if (i < j) { F t = a[i]; a[i] = a[j]; a[j] = t; }
    }
    std::vector<F> tw(n / 2 ? n / 2 : 1);
    if (n >= 2) {
        tw[0] = F::one();
        for (size_t k = 1; k < n / 2; ++k) tw[k] = tw[k - 1] * w;
    }
    (void)threads;
    for (size_t len = 2; len <= n; len <<= 1) {
        size_t half = len / 2, step = n / len;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
#endif
        for (size_t blk = 0; blk < n / len; ++blk) {
            size_t start = blk * len;
            for (size_t k = 0; k < half; ++k) {
                F u = a[start + k];
                F t = a[start + k + half] * tw[k * step];
                a[start + k] = u + t;
                a[start + k + half] = u - t;
            }
        }
    }
    if (inverse) {
        F ninv = F::from_u64((uint64_t)n).inv();
        for (size_t i = 0; i < n; ++i) a[i] = a[i] * ninv;
    }
    for (size_t i = 0; i < n; ++i) a[i].to_canonical(data + 4 * i);
    return 0;
}

// ------------------------------------------------------------------------------------------
// dispatch helpers
// ------------------------------------------------------------------------------------------

typedef Fp1<BnFqTag> BnG1K;
typedef Fp2<BnFqTag> BnG2K;
typedef Fp1<BlsFqTag> BlsG1K;
typedef Fp2<BlsFqTag> BlsG2K;

#define DISPATCH_GROUP(curve, group, CALL)                              \
    if ((curve) == 0 && (group) == 1) { CALL(BnG1K, BnFrTag) }          \
    else if ((curve) == 0 && (group) == 2) { CALL(BnG2K, BnFrTag) }     \
    else if ((curve) == 1 && (group) == 1) { CALL(BlsG1K, BlsFrTag) }   \
    else if ((curve) == 1 && (group) == 2) { CALL(BlsG2K, BlsFrTag) }   \
    else return 4;

// ONE output of a transform straight from the definition (what src/bn254/polynomial.rs:535-571 computes
// through ark-poly): X_k = sum_j a_j w^(jk), the inverse with w^-1 and the factor 1/n.  O(n) per output: the checker for sizes
// where the full transform above is too slow to run inside a test (2^28, the largest BN254 domain)
template <class FrTag>
static int ntt_spot(const uint64_t* data, int log_n, int inverse, uint64_t k, int two_adicity, uint64_t gen, int threads, uint64_t* out) {
    typedef Fp<FrTag> F;
    if (log_n > two_adicity) return 2;
    const size_t n = (size_t)1 << log_n;
    F w = root_of_unity<FrTag>(log_n, two_adicity, gen);
    if (inverse) w = w.inv();
    uint64_t e[1] = {k & (n - 1)};
    const F x = w.pow(e, 1);
    const size_t len = n < 65536 ? n : 65536, chunks = n / len;
    uint64_t le[1] = {len};
    const F x_len = x.pow(le, 1);
    std::vector<F> part(chunks);
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
    for (size_t c = 0; c < chunks; ++c) {
        F acc = F::zero();
        for (size_t j = len; j-- > 0;) acc = acc * x + F::from_canonical(data + 4 * (c * len + j));
        part[c] = acc;
    }
    F total = F::zero();
    for (size_t c = chunks; c-- > 0;) total = total * x_len + part[c];
    if (inverse) total = total * F::from_u64(n).inv();
    total.to_canonical(out);
    return 0;
}

extern "C" {

// curve: 0 = BN254, 1 = BLS12-381.  group: 1 = G1, 2 = G2.

int orc_ark_window(uint64_t n) { return ark_window((size_t)n); }

int orc_msm(int curve, int group, uint64_t n, const uint64_t* scalars, const uint64_t* bases, uint64_t* out,
            int threads, int c_override) {
#define CALL(K, FR) store_aff<K>(out, pippenger<K, FR>((size_t)n, scalars, bases, threads, c_override).to_affine());
    DISPATCH_GROUP(curve, group, CALL)
#undef CALL
    return 0;
}

// out[i] = scalars[i] * bases[i]  (bases_stride 0 => one fixed base)
int orc_batch_mul(int curve, int group, uint64_t n, const uint64_t* scalars, const uint64_t* bases, int fixed_base,
                  uint64_t* out, int threads) {
    (void)threads;
#define CALL(K, FR)                                                                         \
    {                                                                                       \
        const size_t pw = 2 * K::WORDS;                                                     \
        _Pragma("omp parallel for schedule(dynamic, 64) num_threads(threads > 0 ? threads : 1)") \
        for (uint64_t i = 0; i < n; ++i) {                                                  \
            uint64_t s[4];                                                                  \
            reduce_scalar<FR>(s, scalars + 4 * i);                                          \
            Aff<K> b = load_aff<K>(bases + (fixed_base ? 0 : i * pw));                      \
            store_aff<K>(out + i * pw, scalar_mul<K>(b, s).to_affine());                    \
        }                                                                                   \
    }
    DISPATCH_GROUP(curve, group, CALL)
#undef CALL
    return 0;
}

int orc_point_add(int curve, int group, const uint64_t* a, const uint64_t* b, uint64_t* out) {
#define CALL(K, FR)                                         \
    {                                                       \
        Aff<K> pa = load_aff<K>(a), pb = load_aff<K>(b);    \
        Jac<K> j = Jac<K>::infinity().add_affine(pa).add_affine(pb); \
        store_aff<K>(out, j.to_affine());                   \
    }
    DISPATCH_GROUP(curve, group, CALL)
#undef CALL
    return 0;
}

int orc_ntt_spot(int curve, const uint64_t* data, int log_n, int inverse, uint64_t k, uint64_t* out, int threads) {
    if (curve == 0) return ntt_spot<BnFrTag>(data, log_n, inverse, k, 28, 5, threads, out);
    if (curve == 1) return ntt_spot<BlsFrTag>(data, log_n, inverse, k, 32, 7, threads, out);
    return 4;
}

int orc_ntt(int curve, uint64_t* data, int log_n, int inverse, int threads) {
    if (curve == 0) return ntt_inplace<BnFrTag>(data, log_n, inverse, 28, 5, threads);
    if (curve == 1) return ntt_inplace<BlsFrTag>(data, log_n, inverse, 32, 7, threads);
    return 4;
}

// sum_i a_i b_i mod r: the discrete logarithm of MSM(a, b_i G), the closed-form expectation of SURVEY.md 8c(2) at sizes
// where Python's big integers are too slow (2^26 pairs)
int orc_dot(int curve, uint64_t n, const uint64_t* a, const uint64_t* b, uint64_t* out, int threads) {
#define BODY(FR)                                                                                  \
    {                                                                                             \
        typedef Fp<FR> F;                                                                         \
        const int T = threads > 0 ? threads : 1;                                                  \
        std::vector<F> part(T, F::zero());                                                        \
        _Pragma("omp parallel for schedule(static) num_threads(T)")                               \
        for (int t = 0; t < T; ++t) {                                                             \
            F acc = F::zero();                                                                    \
            for (uint64_t i = n * t / T; i < n * (t + 1) / T; ++i)                                \
                acc = acc + F::from_canonical(a + 4 * i) * F::from_canonical(b + 4 * i);          \
            part[t] = acc;                                                                        \
        }                                                                                         \
        F total = F::zero();                                                                      \
        for (int t = 0; t < T; ++t) total = total + part[t];                                      \
        total.to_canonical(out);                                                                  \
    }
    if (curve == 0) BODY(BnFrTag) else if (curve == 1) BODY(BlsFrTag) else return 4;
#undef BODY
    return 0;
}

// op: 0 = mul, 1 = add, 2 = sub (element-wise over Fr)
int orc_vec_op(int curve, int op, uint64_t n, const uint64_t* a, const uint64_t* b, uint64_t* out) {
#define BODY(FR)                                                              \
    for (uint64_t i = 0; i < n; ++i) {                                        \
        Fp<FR> x = Fp<FR>::from_canonical(a + 4 * i), y = Fp<FR>::from_canonical(b + 4 * i); \
        Fp<FR> z = op == 0 ? x * y : (op == 1 ? x + y : x - y);               \
        z.to_canonical(out + 4 * i);                                          \
    }
    if (curve == 0) { BODY(BnFrTag) } else if (curve == 1) { BODY(BlsFrTag) } else return 4;
#undef BODY
    return 0;
}

// QAP witness polynomials (python/zksnake/groth16/qap.py:57-69) from the three evaluation vectors
// a = A.w, b = B.w, c = C.w (length n = 2^log_n): writes u, v (n coeffs each), h (n coeffs; h[n-1] = 0)
// returns 0 on success, 1 when (u*v - w) is not divisible by X^n - 1.
int orc_qap_h(int curve, int log_n, const uint64_t* a, const uint64_t* b, const uint64_t* c,
              uint64_t* u, uint64_t* v, uint64_t* h, int threads) {
    const size_t n = (size_t)1 << log_n;
    std::vector<uint64_t> U(8 * n, 0), V(8 * n, 0), W(4 * n);
    memcpy(U.data(), a, 32 * n);
    memcpy(V.data(), b, 32 * n);
    memcpy(W.data(), c, 32 * n);
    int rc = 0;
    rc |= orc_ntt(curve, U.data(), log_n, 1, threads);
    rc |= orc_ntt(curve, V.data(), log_n, 1, threads);
    rc |= orc_ntt(curve, W.data(), log_n, 1, threads);
    if (rc) return rc;
    memcpy(u, U.data(), 32 * n);
    memcpy(v, V.data(), 32 * n);
    rc |= orc_ntt(curve, U.data(), log_n + 1, 0, threads);
    rc |= orc_ntt(curve, V.data(), log_n + 1, 0, threads);
    if (rc) return rc;
    orc_vec_op(curve, 0, 2 * n, U.data(), V.data(), U.data());
    rc |= orc_ntt(curve, U.data(), log_n + 1, 1, threads);
    if (rc) return rc;
    // hz = uv - w ; divide by X^n - 1: q = high half, rem = low + high
    std::vector<uint64_t> lo(4 * n);
    orc_vec_op(curve, 2, n, U.data(), W.data(), lo.data());          // low half minus w
    orc_vec_op(curve, 1, n, lo.data(), U.data() + 4 * n, lo.data()); // + high half = remainder
    for (size_t i = 0; i < 4 * n; ++i) if (lo[i]) return 1;
    memcpy(h, U.data() + 4 * n, 32 * n);
    return 0;
}

int orc_max_threads() {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

}  // extern "C"
