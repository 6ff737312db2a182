/* pyints.c -- CPython helper of the host shim: Python ints <-> little-endian 64-bit limb arrays.
 *
 * The reference marshals every scalar through pyo3's BigUint extraction (`Fr::from(BigUint)`, src/bn254/curve.rs:358-361;
 * negative ints raise OverflowError there).  The reference-shaped API of this package (prove(list[int], list[int])) has to
 * do the same for 2^20 and more witness values per call; `int.to_bytes` per element costs ~180 ms at 2^20, this loop ~40.
 * Pure marshalling: no field arithmetic happens here (values >= the modulus take the Python-level `%`).
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>
#include <string.h>

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
    int have_mod = modulus != Py_None;
    PyObject* result = NULL;
    if ((size_t)out.len < (size_t)n * nbytes || nbytes > sizeof(mod_bytes)) {
        PyErr_SetString(PyExc_ValueError, "output buffer too small");
        goto done;
    }
    if (have_mod && _PyLong_AsByteArray((PyLongObject*)modulus, mod_bytes, nbytes, 1, 0) < 0) goto done;
    for (Py_ssize_t i = 0; i < n; ++i) {
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
            PyObject* r = PyNumber_Remainder(v, modulus);
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
