/* forwarder: the functional mini OpenCV of the host-reference harness (test infrastructure) */
#include "../cv_mini.hpp"
