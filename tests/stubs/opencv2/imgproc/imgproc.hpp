// test infrastructure: see ../opencv_stub.hpp
#include "../opencv_stub.hpp"
