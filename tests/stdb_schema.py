"""SURVEY.md Appendix A (the STdb wire schema implied by STAN_Database's protobuf-net attributes:
Database.cs:12-21, Node.cs:11-21, Element.cs:14-23, MatrixST.cs:17-19, Material.cs:9-14,
BoundaryCondition.cs:10-14, Analysis.cs:8-13, Information.cs:9,35-40) as a google.protobuf
DYNAMIC descriptor: an independent, third-party implementation of the protobuf wire format that
the hand-written codec of libstan_host.so (stan_amd/host/stdb.cpp) is pinned against.

TEST INFRASTRUCTURE ONLY.  Dictionaries are modelled as `repeated Entry {key = 1; value = 2}`
(wire-identical to a proto map and to protobuf-net's dictionary encoding) so that the
third-party decoder preserves the wire order of the entries -- Database.AssignDOF depends on it
("Semantics the solver relies on").  Scalar presence follows proto3 (zero
defaults are not written), which is protobuf-net's implicit-zero-default rule."""
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

T = descriptor_pb2.FieldDescriptorProto


def _field(msg, name, number, ftype, repeated=False, type_name=None, packed=None):
    f = msg.field.add()
    f.name, f.number, f.type = name, number, ftype
    f.label = T.LABEL_REPEATED if repeated else T.LABEL_OPTIONAL
    if type_name:
        f.type_name = ".stan." + type_name
    if packed is not None:
        f.options.packed = packed
    return f


def build(packed):
    """Message classes of the STdb schema; `packed` selects the encoding of repeated scalars
    (protobuf-net writes them unpacked unless IsPacked is set; the GUI's files are unpacked)."""
    fd = descriptor_pb2.FileDescriptorProto()
    fd.name = "stan_stdb_%s.proto" % ("packed" if packed else "unpacked")
    fd.package = "stan"
    fd.syntax = "proto3"

    def msg(name):
        m = fd.message_type.add()
        m.name = name
        return m

    def entry(name, value_type):
        m = msg(name)
        _field(m, "key", 1, T.TYPE_INT32)
        _field(m, "value", 2, T.TYPE_MESSAGE, type_name=value_type)

    m = msg("MatrixST")
    _field(m, "M", 1, T.TYPE_DOUBLE, True, packed=packed)
    _field(m, "Rows", 2, T.TYPE_INT32)
    _field(m, "Cols", 3, T.TYPE_INT32)

    m = msg("Node")
    _field(m, "ID", 1, T.TYPE_INT32)
    for i, c in enumerate("XYZ"):
        _field(m, c, 2 + i, T.TYPE_DOUBLE)
    _field(m, "EList", 5, T.TYPE_INT32, True, packed=packed)
    _field(m, "DOF", 6, T.TYPE_INT32, True, packed=packed)
    for i, c in enumerate("XYZ"):
        _field(m, "Disp" + c, 7 + i, T.TYPE_DOUBLE, True, packed=packed)

    m = msg("Element")
    _field(m, "ID", 1, T.TYPE_INT32)
    _field(m, "Type", 2, T.TYPE_STRING)
    _field(m, "PID", 3, T.TYPE_INT32)
    _field(m, "MatID", 4, T.TYPE_INT32)
    _field(m, "NList", 5, T.TYPE_INT32, True, packed=packed)
    _field(m, "Strain", 6, T.TYPE_MESSAGE, True, "MatrixST")
    _field(m, "Stress", 7, T.TYPE_MESSAGE, True, "MatrixST")

    m = msg("Material")
    _field(m, "ID", 1, T.TYPE_INT32)
    _field(m, "Type", 2, T.TYPE_STRING)
    _field(m, "Name", 3, T.TYPE_STRING)
    _field(m, "E", 4, T.TYPE_DOUBLE)
    _field(m, "Poisson", 5, T.TYPE_DOUBLE)
    _field(m, "ColorID", 6, T.TYPE_INT32)

    entry("MatrixEntry", "MatrixST")
    m = msg("BoundaryCondition")
    _field(m, "Type", 1, T.TYPE_STRING)
    _field(m, "Name", 2, T.TYPE_STRING)
    _field(m, "ID", 3, T.TYPE_INT32)
    _field(m, "NodalValues", 4, T.TYPE_MESSAGE, True, "MatrixEntry")
    _field(m, "ColorID", 5, T.TYPE_INT32)

    m = msg("Analysis")
    _field(m, "Type", 1, T.TYPE_STRING)
    _field(m, "LinSolver", 2, T.TYPE_STRING)
    _field(m, "LinSolverTolerance", 3, T.TYPE_DOUBLE)
    _field(m, "LinSolverIterMax", 4, T.TYPE_INT32)
    _field(m, "IncNumb", 5, T.TYPE_INT32)
    _field(m, "Result_StepNo", 6, T.TYPE_INT32)

    m = msg("PartInfo")
    _field(m, "ColorID", 1, T.TYPE_INT32)
    _field(m, "MatID", 2, T.TYPE_INT32)
    _field(m, "Name", 3, T.TYPE_STRING)
    _field(m, "HEX_Type", 4, T.TYPE_STRING)
    _field(m, "PENTA_Type", 5, T.TYPE_STRING)
    _field(m, "TET_Type", 6, T.TYPE_STRING)
    entry("PartInfoEntry", "PartInfo")
    m = msg("Information")
    _field(m, "InfoPart", 1, T.TYPE_MESSAGE, True, "PartInfoEntry")

    entry("NodeEntry", "Node")
    entry("ElementEntry", "Element")
    entry("MaterialEntry", "Material")
    entry("BCEntry", "BoundaryCondition")
    m = msg("Database")
    _field(m, "NodeLib", 1, T.TYPE_MESSAGE, True, "NodeEntry")
    _field(m, "ElemLib", 2, T.TYPE_MESSAGE, True, "ElementEntry")
    _field(m, "MatLib", 3, T.TYPE_MESSAGE, True, "MaterialEntry")
    _field(m, "BCLib", 4, T.TYPE_MESSAGE, True, "BCEntry")
    _field(m, "nDOF", 5, T.TYPE_INT32)
    _field(m, "AnalysisLib", 6, T.TYPE_MESSAGE, type_name="Analysis")
    _field(m, "Info", 7, T.TYPE_MESSAGE, type_name="Information")

    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    return message_factory.GetMessageClass(pool.FindMessageTypeByName("stan.Database"))
