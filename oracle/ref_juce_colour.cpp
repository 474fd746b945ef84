// ref_juce_colour.cpp -- the ONE link of the reference's colour chain that compiles in this image from the reference's own sources:
// juce::Colour::withRotatedHue (JuceLibraryCode/modules/juce_graphics/colour/juce_Colour.cpp:33-107, :331-336), which
// Signalizer::ColourRotation::operator[] (Source/Common/CommonSignalizer.h:931-937) calls to turn the spectrum colours by pair
// (TransformConstant::generateSpectrogramColourRotation, Source/Spectrum/TransformConstant.h:55-65).
//
// Test infrastructure, container-only: the reference's files are #included WHERE THEY LIE under /root/reference (nothing of them is
// copied into this repository), the output goes to oracle/_ref/libjuce_colour_ref.so (git-ignored; see oracle/Makefile `_ref`).
// juce_core.h is the reference's own header; juce::String's out-of-line members stay unresolved in the shared object and are never
// called by anything below (link with -z lazy, load with RTLD_LAZY).  Nothing here stands in for a header, library or tool the image lacks.
#include <cstddef>
#include <cstdint>

#include "/root/reference/JuceLibraryCode/modules/juce_core/juce_core.h"

namespace juce {
#include "/root/reference/JuceLibraryCode/modules/juce_graphics/colour/juce_PixelFormats.h"
#include "/root/reference/JuceLibraryCode/modules/juce_graphics/colour/juce_Colour.h"
#include "/root/reference/JuceLibraryCode/modules/juce_graphics/colour/juce_Colours.h"
#include "/root/reference/JuceLibraryCode/modules/juce_graphics/colour/juce_Colour.cpp"
#include "/root/reference/JuceLibraryCode/modules/juce_graphics/colour/juce_Colours.cpp"
}

extern "C" {

// juce::Colour(r, g, b).withRotatedHue(amount) -> rgb
void sgzref_rotate_hue_rgb8(const uint8_t rgb[3], float amount, uint8_t out[3])
{
    const juce::Colour c = juce::Colour(rgb[0], rgb[1], rgb[2]).withRotatedHue(amount);
    out[0] = c.getRed(); out[1] = c.getGreen(); out[2] = c.getBlue();
}

// ColourRotation(base, size, stereo)[index] (CommonSignalizer.h:931-937: `if (stereo) index &= ~0x1ull; return
// base.withRotatedHue(index / size);` with `float size`): the two statements restated around the compiled withRotatedHue
void sgzref_colour_rotation_rgb8(const uint8_t base[3], uint64_t index, float size, int stereo, uint8_t out[3])
{
    if (stereo) index &= ~0x1ull;
    const juce::Colour c = juce::Colour(base[0], base[1], base[2]).withRotatedHue(index / size);
    out[0] = c.getRed(); out[1] = c.getGreen(); out[2] = c.getBlue();
}

}
