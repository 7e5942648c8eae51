# cuBVH.jl -- replaces src/cuBVH.jl (CuBVHTree, cu(::BVHTree), src/cuBVH.jl:4-22) and the Adapt rule for trees of meshes
# (src/mesh.jl:24-43): instead of copying a BVHTree to the device node by node and object by object, the tree is written
# down as the arrays an RtBvh record points to -- node ids stay 1-based exactly as BVHNode.left / .right hold them
# (src/BVH.jl:5-10) -- and the library builds its own device structure from them (rt_scene_create).
include("BVH.jl")          # the reference's own file (src/BVH.jl): BVHNode, BVHTree, build_bvh, hit
include("librt_amd.jl")

# What the library needs of one BVHTree, kept alive together (the RtBvh record holds raw pointers into these arrays).
struct FlatBvh{F}
    kind::Int32                 # 0: generic tree (src/BVH.jl), 1: tree over meshes (src/mesh.jl:250-305)
    objects::Vector{Int32}      # 0-based indices into the object table, in tree order (tree.objects after the build)
    bounds::Vector{F}           # 6 per node slot: min(3), max(3), in the tree's element type (BVHNode{F}, src/BVH.jl:5-10)
    left::Vector{Int32}
    right::Vector{Int32}
    leaf::Vector{UInt8}         # 0 interior, 1 leaf, 255 slot not reached from node 1
end

# Walks the tree from node 1 and copies the nodes that are reached: the node vector of a tree over meshes is allocated
# for the worst case (4 * triangles - 1 slots, src/mesh.jl:300) and most of it stays uninitialised.
function flat_bvh(tree::BVHTree{F,T}, object_index::Function) where {F,T}
    n = length(tree.nodes)
    kind = T <: Mesh ? Int32(1) : Int32(0)      # (Mesh: src/mesh.jl:5-9, defined by the time a tree is flattened)
    flat = FlatBvh{F}(kind, Int32[object_index(o) for o in tree.objects], zeros(F, 6n), zeros(Int32, n), zeros(Int32, n),
                   fill(0xff, n))
    n == 0 && return flat
    stack = Tuple{Int,Bool}[(1, kind == 1)]     # (node id, the node belongs to the level of the meshes)
    while !isempty(stack)
        id, mesh_level = pop!(stack)
        node = tree.nodes[id]
        b = node.box
        flat.bounds[6id-5:6id] .= (F(b.min.x), F(b.min.y), F(b.min.z), F(b.max.x), F(b.max.y), F(b.max.z))
        flat.left[id] = Int32(node.left)
        flat.right[id] = Int32(node.right)
        flat.leaf[id] = node.is_leaf ? 0x01 : 0x00
        if !node.is_leaf
            push!(stack, (node.left, mesh_level), (node.right, mesh_level))
        elseif mesh_level
            # a leaf of the mesh level: left = root of that mesh's triangle tree, right = position of the mesh
            # (src/mesh.jl:253-257); its triangle tree follows in the same node vector
            push!(stack, (node.left, false))
        end
    end
    return flat
end

rt_bvh(flat::FlatBvh{F}) where F = records(F).bvh(flat.kind, Int32(length(flat.objects)), pointer(flat.objects), Int32(length(flat.leaf)), Int32(0),
                                                  pointer(flat.bounds), pointer(flat.left), pointer(flat.right), pointer(flat.leaf))
element_type(::BVHTree{F,T}) where {F,T} = F
