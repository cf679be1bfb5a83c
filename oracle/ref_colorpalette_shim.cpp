// TEST INFRASTRUCTURE ONLY -- never linked into the product library.
//
// extern "C" driver around the reference's own CColorPalette, compiled *as is*
// from the read-only mount (/root/reference/CColorpalette.cpp + CColorpalette.h
// + ColormapData.h; no other dependency, no stand-in headers).  The recipe that
// builds it is oracle/Makefile (target `ref`), output oracle/_ref/ (git-ignored).
// It pins oracle/jsg_oracle.py's colour-map restatement and generates the golden
// vectors under tests/golden/ (oracle/gen_golden.py).
//
// The class keeps its table `protected` (CColorpalette.h:51-58), hence the
// subclass.
#include "CColorpalette.h"

namespace {
struct Probe : public CColorPalette {
    Probe(int n, int scheme) : CColorPalette(n, scheme) {}
    const std::vector<int>& table() const { return m_Color; }
    float mult() const { return m_AccessMult; }
    float vmin() const { return m_Min; }
    float vmax() const { return m_Max; }
    int index_of(float value) const {  // same arithmetic as getRGBColor, returns the index
        if (value >= m_Max) value = m_Max * 0.9999f;
        if (value < m_Min) value = m_Min;
        int index = int((value - m_Min) * m_AccessMult);
        return index < m_NrOfColors ? index : m_NrOfColors - 1;
    }
};
}  // namespace

extern "C" {

// CColorPalette(n, scheme) -> table of n ints (0x00RRGGBB)
int ref_cp_lut(int n_colors, int scheme, int* out) {
    Probe p(n_colors, scheme);
    for (int i = 0; i < n_colors; ++i) out[i] = p.table()[i];
    return 0;
}

// setValueRange(lo, hi) then getRGBColor(v[i]) for every i; also reports the state
int ref_cp_map(int n_colors, int scheme, float lo, float hi, const float* v, int n,
               int* rgb_out, int* idx_out, float* state3 /* min,max,mult */) {
    Probe p(n_colors, scheme);
    p.setValueRange(lo, hi);
    for (int i = 0; i < n; ++i) {
        rgb_out[i] = p.getRGBColor(v[i]);
        if (idx_out) idx_out[i] = p.index_of(v[i]);
    }
    if (state3) { state3[0] = p.vmin(); state3[1] = p.vmax(); state3[2] = p.mult(); }
    return 0;
}

// setColorSceme on a live object (the GUI path, Spectrogram.cpp:400)
int ref_cp_switch_scheme(int n_colors, int scheme_a, int scheme_b, int* out) {
    Probe p(n_colors, scheme_a);
    p.setColorSceme(scheme_b);
    for (int i = 0; i < n_colors; ++i) out[i] = p.table()[i];
    return 0;
}
}
