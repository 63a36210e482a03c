// ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path (see vec3.hpp).
// BVHModel<OBBRSS> traversal restatement -- filled in below.
#pragma once
