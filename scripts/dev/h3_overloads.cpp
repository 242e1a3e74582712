// SURVEY hazard H3, narrowed: with the STANDARD headers of include/kazen/common.h:3-15 and nothing else, which overloads do the reference's unqualified cos / sin / pow / abs / sqrt / atan / log /
// exp / tan / acos (ggx_brdf.h:23,108-109,143,166; warp.cpp:122-129) find inside a namespace?   g++ -std=c++17 h3_overloads.cpp && ./a.out
// g++ 11.4 / libstdc++ and clang 22 (this image): every one resolves to the C library's DOUBLE function, and abs(float) to ::abs(int). The float overloads (std::cos(float), std::abs(float), ...)
// reach the global namespace only through libstdc++'s <math.h> / <stdlib.h> wrappers (`using std::cos;` ...), i.e. only if one of the reference's third-party headers includes those C headers:
// Eigen's <emmintrin.h> -> <mm_malloc.h> -> <stdlib.h> does (so abs(float) is the float one - with ::abs(int) the published pictures would be black where ggx_brdf.h takes abs of a cosine);
// whether anything includes <math.h> depends on the Imath / OpenImageIO versions the reference was built against, which are not pinned. The oracle and the HIP path keep SURVEY's reading (float
// functions); under the other reading warp.cpp:122-129 and ggx_brdf.h:108-109 would evaluate in double and narrow once - last-ulp differences that no published artefact can decide.
#include <string>
#include <vector>
#include <map>
#include <type_traits>
#include <iostream>
#include <algorithm>
#include <limits>
#include <stdint.h>
#include <cstring>
#include <cmath>
#include <cassert>
namespace kazen {
template <class T> const char *nm() { return std::is_same<T, float>::value ? "float" : std::is_same<T, double>::value ? "double" : std::is_same<T, int>::value ? "int" : "other"; }
void show() {
    float x = 0.3f;
    std::cout << "cos(float) -> " << nm<decltype(cos(x))>() << "\n";
    std::cout << "sin(float) -> " << nm<decltype(sin(x))>() << "\n";
    std::cout << "pow(float, float) -> " << nm<decltype(pow(x, 5.0f))>() << "\n";
    std::cout << "pow(float, int) -> " << nm<decltype(pow(x, 2))>() << "\n";
    std::cout << "abs(float) -> " << nm<decltype(abs(x))>() << "\n";
    std::cout << "sqrt(float) -> " << nm<decltype(sqrt(x))>() << "\n";
    std::cout << "atan(float) -> " << nm<decltype(atan(x))>() << "\n";
    std::cout << "log(float) -> " << nm<decltype(log(x))>() << "\n";
    std::cout << "exp(float) -> " << nm<decltype(exp(x))>() << "\n";
    std::cout << "tan(float) -> " << nm<decltype(tan(x))>() << "\n";
    std::cout << "acos(float) -> " << nm<decltype(acos(x))>() << "\n";
}
}
int main() { kazen::show(); }
