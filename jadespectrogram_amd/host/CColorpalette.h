// Drop-in mirror of the reference's CColorPalette (reference CColorpalette.h:6-61) on top of the C-ABI: the table
// and the value range come from libjsg.so (jsg_colormap_build / jsg_colormap_range, bit-identical to the reference's
// ComputeColors / setValueRange); getRGBColor stays an inline host accessor for single values (colour bar, labels).
// Bulk pixel mapping is done on the GPU (jsg_colormap_launch / jsg_display_update).
#pragma once
#include <cstdint>
#include <vector>

#include "../../include/jsg.h"

class CColorPalette {
public:
    enum { kMono = 0, kBW, kHot, kRainbow, kViridis, kPlasma, kJade };

    CColorPalette() : CColorPalette(2, kMono) {}
    explicit CColorPalette(int NrOfColors) : CColorPalette(NrOfColors, kMono) {}
    CColorPalette(int NrOfColors, int ColorScheme)
        : m_NrOfColors(NrOfColors), m_Max(1.f), m_Min(0.f), m_ColorScheme(ColorScheme), m_InvertScheme(0) {
        m_AccessMult = float(m_NrOfColors) / (m_Max - m_Min);
        AllocateColors();
    }
    ~CColorPalette() = default;

    void setValueRange(float Min, float Max) { jsg_colormap_range(m_NrOfColors, Min, Max, &m_Min, &m_Max, &m_AccessMult); }
    void setNrOfColors(int NrOfColors) {
        m_NrOfColors = NrOfColors;
        m_AccessMult = float(m_NrOfColors) / (m_Max - m_Min);
        AllocateColors();
    }
    void setColorSceme(int ColorScheme) {
        m_ColorScheme = ColorScheme;
        ComputeColors();
    }
    void setInvertStatus(bool status) { m_InvertScheme = status; }   // never used by the plugin; not applied

    inline int getRGBColor(float value) const {
        if (value >= m_Max) value = m_Max * 0.9999f;
        if (value < m_Min) value = m_Min;
        const int index = int((value - m_Min) * m_AccessMult);
        return index < m_NrOfColors ? m_Color[size_t(index)] : m_Color[size_t(m_NrOfColors - 1)];
    }
    float getValue(int iColor) const {
        for (int kk = 0; kk < m_NrOfColors; ++kk)
            if (m_Color[size_t(kk)] == iColor) return float(kk) / m_AccessMult + m_Min;
        return 100000000000000000000000000000.f;
    }

    // accessors used by the GPU display path
    const std::vector<int>& table() const { return m_Color; }
    int nrOfColors() const { return m_NrOfColors; }
    int colorScheme() const { return m_ColorScheme; }
    float minValue() const { return m_Min; }
    float maxValue() const { return m_Max; }

protected:
    void ComputeColors() {
        static_assert(sizeof(int) == sizeof(int32_t), "int is 32 bits");
        jsg_colormap_build(m_NrOfColors, m_ColorScheme, reinterpret_cast<int32_t*>(m_Color.data()));
    }
    void AllocateColors() {
        m_Color.resize(size_t(m_NrOfColors));
        ComputeColors();
    }
    std::vector<int> m_Color;
    int m_NrOfColors;
    float m_Max;
    float m_Min;
    float m_AccessMult;
    int m_ColorScheme;
    int m_InvertScheme;
};
