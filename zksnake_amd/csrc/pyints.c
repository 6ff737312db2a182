/* pyints.c -- CPython helper of the host shim: Python ints <-> little-endian 64-bit limb arrays.
 *
 * The reference marshals every scalar through pyo3's BigUint extraction (`Fr::from(BigUint)`, src/bn254/curve.rs:358-361;
 * negative ints raise OverflowError there).  The reference-shaped API of this package (prove(list[int], list[int])) has to
 * do the same for 2^20 and more witness values per call; `int.to_bytes` per element costs ~180 ms at 2^20, one thread of this loop ~45, eight ~8.
 * Pure marshalling: no field arithmetic happens here (values >= the modulus take the Python-level `%`).
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

/* value of a non-negative PyLong into `words` little-endian 64-bit limbs, straight from its 30-bit digits (CPython 3.8 .. 3.11
 * layout; anything else goes through _PyLong_AsByteArray).  Returns 0 ok, 1 does not fit, -1 not applicable. */
static int pack_digits(PyObject* v, unsigned char* dst, size_t words) {
#if PY_VERSION_HEX < 0x030C0000 && PYLONG_BITS_IN_DIGIT == 30
    const PyLongObject* lv = (const PyLongObject*)v;
    const Py_ssize_t nd = Py_SIZE(lv);
    uint64_t out[16];
    if (nd < 0 || words > 16) return -1;
    for (size_t w = 0; w < words; ++w) out[w] = 0;
    size_t bit = 0;
    for (Py_ssize_t i = 0; i < nd; ++i, bit += 30) {
        const uint64_t d = lv->ob_digit[i];
        const size_t w = bit >> 6, o = bit & 63;
        if (w >= words) {
            if (d) return 1;
            continue;
        }
        out[w] |= d << o;
        if (o > 34) {
            const uint64_t hi = d >> (64 - o);
            if (w + 1 < words) out[w + 1] |= hi;
            else if (hi) return 1;
        }
    }
    memcpy(dst, out, words * 8);
    return 0;
#else
    (void)v; (void)dst; (void)words;
    return -1;
#endif
}

/* ---- the bulk of a witness: exact, non-negative ints that fit and are below the modulus -------------------------------
 * Worker threads repack those straight from the digits while the calling thread keeps the GIL (so nothing can mutate the
 * sequence or its ints meanwhile; the workers only read, they touch no reference count).  Everything else -- other
 * types, negative or oversized values, values >= the modulus -- is flagged and handled afterwards in index order by the
 * general path below, so the first offending element still decides the exception. */
typedef struct {
    PyObject** items;
    Py_ssize_t begin, end;
    size_t words;
    unsigned char* dst;
    const uint64_t* mod;   /* NULL = no modulus */
    unsigned char* slow;   /* per element: 1 = leave to the general path */
} PackJob;

static void* pack_worker(void* arg) {
    PackJob* job = (PackJob*)arg;
    const size_t nbytes = job->words * 8;
    for (Py_ssize_t i = job->begin; i < job->end; ++i) {
        if (i + 8 < job->end) __builtin_prefetch(job->items[i + 8]);
        PyObject* item = job->items[i];
        unsigned char* p = job->dst + (size_t)i * nbytes;
        int ok = PyLong_CheckExact(item) && pack_digits(item, p, job->words) == 0;
        if (ok && job->mod) {
            uint64_t v[16];
            memcpy(v, p, nbytes);
            int cmp = 0;
            for (Py_ssize_t k = (Py_ssize_t)job->words - 1; k >= 0 && cmp == 0; --k) cmp = v[k] > job->mod[k] ? 1 : (v[k] < job->mod[k] ? -1 : 0);
            ok = cmp < 0;
        }
        job->slow[i] = (unsigned char)!ok;
    }
    return NULL;
}

#define PACK_MAX_THREADS 8
#define PACK_PARALLEL_MIN 16384

static void pack_fast(PyObject** items, Py_ssize_t n, size_t words, unsigned char* dst, const uint64_t* mod, unsigned char* slow) {
    long cores = sysconf(_SC_NPROCESSORS_ONLN);
    int threads = n >= PACK_PARALLEL_MIN ? (int)(cores > PACK_MAX_THREADS ? PACK_MAX_THREADS : (cores < 1 ? 1 : cores)) : 1;
    PackJob jobs[PACK_MAX_THREADS];
    pthread_t tids[PACK_MAX_THREADS];
    int started[PACK_MAX_THREADS];
    const Py_ssize_t chunk = (n + threads - 1) / threads;
    for (int t = 0; t < threads; ++t) {
        Py_ssize_t b = t * chunk, e = b + chunk > n ? n : b + chunk;
        if (b > n) b = n;
        jobs[t] = (PackJob){items, b, e, words, dst, mod, slow};
        started[t] = 0;
        if (t > 0) started[t] = pthread_create(&tids[t], NULL, pack_worker, &jobs[t]) == 0;
    }
    pack_worker(&jobs[0]);
    for (int t = 1; t < threads; ++t) {
        if (started[t]) pthread_join(tids[t], NULL);
        else pack_worker(&jobs[t]);
    }
}

/* ints_to_limbs(seq, words, modulus_or_None, out) -> None.  out: writable buffer of len(seq) * words * 8 bytes. */
static PyObject* ints_to_limbs(PyObject* self, PyObject* args) {
    PyObject *seq, *modulus;
    Py_ssize_t words;
    Py_buffer out;
    if (!PyArg_ParseTuple(args, "OnOw*", &seq, &words, &modulus, &out)) return NULL;
    PyObject* fast = PySequence_Fast(seq, "expected a sequence of ints");
    if (!fast) { PyBuffer_Release(&out); return NULL; }
    const Py_ssize_t n = PySequence_Fast_GET_SIZE(fast);
    const size_t nbytes = (size_t)words * 8;
    unsigned char* dst = (unsigned char*)out.buf;
    unsigned char mod_bytes[128];
    uint64_t mod_words[16];
    int have_mod = modulus != Py_None;
    PyObject* result = NULL;
    PyObject* mod_obj = NULL;   /* the modulus as an exact int (new reference) */
    unsigned char* slow = NULL;
    if ((size_t)out.len < (size_t)n * nbytes || nbytes > sizeof(mod_bytes) || words < 1) {
        PyErr_SetString(PyExc_ValueError, "output buffer too small");
        goto done;
    }
    if (have_mod) {
        /* any integer-like modulus (int, numpy integer) through __index__; a non-integer raises TypeError as the pure-Python
         * path does, and a modulus below 1 is rejected instead of being read as unsigned bytes */
        mod_obj = PyNumber_Index(modulus);
        if (!mod_obj) goto done;
        if (_PyLong_Sign(mod_obj) <= 0) {
            PyErr_SetString(PyExc_ValueError, "modulus must be a positive integer");
            goto done;
        }
        if (_PyLong_AsByteArray((PyLongObject*)mod_obj, mod_bytes, nbytes, 1, 0) < 0) goto done;
    }
    slow = (unsigned char*)malloc(n > 0 ? (size_t)n : 1);
    if (!slow) { PyErr_NoMemory(); goto done; }
    memcpy(mod_words, mod_bytes, have_mod ? nbytes : 0);
    pack_fast(PySequence_Fast_ITEMS(fast), n, (size_t)words, dst, have_mod ? mod_words : NULL, slow);
    for (Py_ssize_t i = 0; i < n; ++i) {
        if (!slow[i]) continue;
        PyObject* item = PySequence_Fast_GET_ITEM(fast, i);
        PyObject* v = PyNumber_Index(item);  /* accepts numpy integers too; new reference */
        if (!v) goto done;
        if (_PyLong_Sign(v) < 0) {
            Py_DECREF(v);
            PyErr_SetString(PyExc_OverflowError, "can't convert negative int to unsigned");
            goto done;
        }
        unsigned char* p = dst + (size_t)i * nbytes;
        int need_reduce = 0;
        int packed = pack_digits(v, p, (size_t)words);
        if (packed == 1 && !have_mod) {
            Py_DECREF(v);
            PyErr_SetString(PyExc_OverflowError, "int too big to convert");
            goto done;
        }
        if (packed == 1) {
            need_reduce = 1;
        } else if (packed < 0 && _PyLong_AsByteArray((PyLongObject*)v, p, nbytes, 1, 0) < 0) {
            if (!have_mod || !PyErr_ExceptionMatches(PyExc_OverflowError)) { Py_DECREF(v); goto done; }
            PyErr_Clear();
            need_reduce = 1;
        } else if (have_mod) {
            /* v >= modulus ?  compare from the most significant byte */
            int cmp = 0;
            for (Py_ssize_t k = (Py_ssize_t)nbytes - 1; k >= 0 && cmp == 0; --k) cmp = (int)p[k] - (int)mod_bytes[k];
            need_reduce = cmp >= 0;
        }
        if (need_reduce) {
            PyObject* r = PyNumber_Remainder(v, mod_obj);
            if (!r) { Py_DECREF(v); goto done; }
            int rc = _PyLong_AsByteArray((PyLongObject*)r, p, nbytes, 1, 0);
            Py_DECREF(r);
            if (rc < 0) { Py_DECREF(v); goto done; }
        }
        Py_DECREF(v);
    }
    Py_INCREF(Py_None);
    result = Py_None;
done:
    free(slow);
    Py_XDECREF(mod_obj);
    Py_DECREF(fast);
    PyBuffer_Release(&out);
    return result;
}

/* limbs_to_ints(buffer, words) -> list[int] */
static PyObject* limbs_to_ints(PyObject* self, PyObject* args) {
    Py_buffer in;
    Py_ssize_t words;
    if (!PyArg_ParseTuple(args, "y*n", &in, &words)) return NULL;
    const size_t nbytes = (size_t)words * 8;
    PyObject* list = NULL;
    if (nbytes == 0 || (size_t)in.len % nbytes) {
        PyErr_SetString(PyExc_ValueError, "buffer length is not a multiple of the element size");
        goto done;
    }
    {
        const Py_ssize_t n = (Py_ssize_t)((size_t)in.len / nbytes);
        list = PyList_New(n);
        if (!list) goto done;
        const unsigned char* src = (const unsigned char*)in.buf;
        for (Py_ssize_t i = 0; i < n; ++i) {
            PyObject* v = _PyLong_FromByteArray(src + (size_t)i * nbytes, nbytes, 1, 0);
            if (!v) { Py_CLEAR(list); goto done; }
            PyList_SET_ITEM(list, i, v);
        }
    }
done:
    PyBuffer_Release(&in);
    return list;
}

static PyMethodDef methods[] = {
    {"ints_to_limbs", ints_to_limbs, METH_VARARGS, "ints_to_limbs(seq, words, modulus_or_None, out_buffer)"},
    {"limbs_to_ints", limbs_to_ints, METH_VARARGS, "limbs_to_ints(buffer, words) -> list of ints"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef module = {PyModuleDef_HEAD_INIT, "_pyints", "Python int <-> limb array marshalling", -1, methods};

PyMODINIT_FUNC PyInit__pyints(void) { return PyModule_Create(&module); }
