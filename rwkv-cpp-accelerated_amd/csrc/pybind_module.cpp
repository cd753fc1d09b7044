// pybind_module.cpp -- the reference's pybind module `rwkv` (bindings/pybind/c_binding.cpp:158-175)
// on top of this repo's include/rwkv.h: same function names, arguments and return shapes, so
// bindings/pybind/binding.py (ModelWrapper) works unchanged with SO_LIB_PATH pointing at it.
//
//   initRwkv() -> handle            loadModel(h, path) -> (n_layers, n_embed)
//   modelForward(h, token)          initState(h) / getState(h) -> [5 x np.float64]
//   initOutput(h) / getOutput(h) -> np.float32[50277]        typicalSample(h, temp, tau) -> int
//
// Deliberate fixes of reference quirks:
//   * initState really zeroes the state forward() uses (the reference re-allocates only the
//     deprecated alias pointers, c_binding.cpp:41-60); getState returns the L*D state (the
//     reference copies 50277 elements from those aliases, c_binding.cpp:104-110).
//   * the tokenizer entry points (initTokenizer / tokenizerEncode / tokenizerDecode) are not part
//     of the forward pass and are provided by the host application's tokenizer, not by this module.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include "rwkv.h"
#include "rwkv_sampler.h"

namespace py = pybind11;

static RWKV *M(std::uintptr_t h) { return reinterpret_cast<RWKV *>(h); }

PYBIND11_MODULE(rwkv, m)
{
    m.def("initRwkv", []() { return reinterpret_cast<std::uintptr_t>(new RWKV()); }, "initRwkv");
    m.def("freeRwkv", [](std::uintptr_t h) { delete M(h); }, "destroy a handle (engine extension)");
    m.def("loadModel", [](std::uintptr_t h, const std::string &filename) {
        M(h)->loadFile(filename);
        return std::make_tuple((int64_t)M(h)->num_layers, (int64_t)M(h)->num_embed);
    }, "load");
    m.def("modelForward", [](std::uintptr_t h, int64_t token) { M(h)->forward((unsigned long long)token); }, "rwkvc");
    m.def("initState", [](std::uintptr_t h) {
        RWKV *r = M(h);
        const size_t n = r->num_layers * r->num_embed * r->maxContext;
        for (size_t i = 0; i < n; i++)
            r->state->statexy[i] = r->state->stateaa[i] = r->state->statebb[i] = r->state->statepp[i] = r->state->statedd[i] = 0;
        if (r->residentState) r->pushState();
    }, "initState");
    m.def("getState", [](std::uintptr_t h) {
        RWKV *r = M(h);
        if (r->residentState) r->pullState();
        const size_t n = r->num_layers * r->num_embed;
        py::list out;
        double *src[5] = {r->state->statexy, r->state->stateaa, r->state->statebb, r->state->statepp, r->state->statedd};
        for (int s = 0; s < 5; s++) {
            py::array_t<double> a(n);
            std::copy(src[s], src[s] + n, a.mutable_data());
            out.append(a);
        }
        return out;
    }, "getRwkvState");
    m.def("initOutput", [](std::uintptr_t h) { std::fill(M(h)->out, M(h)->out + 50277, 0.f); }, "initOutput");
    m.def("getOutput", [](std::uintptr_t h) {
        py::array_t<float> a(50277);
        std::copy(M(h)->out, M(h)->out + 50277, a.mutable_data());
        return a;
    }, "getRwkvOutput");
    m.def("typicalSample", [](std::uintptr_t h, float temp, float tau) { return typical(M(h)->out, temp, tau); },
          "typicalSample", py::arg("rwkvp"), py::arg("temp") = 0.9f, py::arg("tau") = 0.8f);
    m.def("setResident", [](std::uintptr_t h, bool on) { M(h)->residentState = on; }, "keep state on the device (engine extension)");
    m.def("decodeGreedy", [](std::uintptr_t h, int64_t first, int64_t n) { return M(h)->decodeGreedy(first, n); },
          "device-side greedy continuation (engine extension)");
}
